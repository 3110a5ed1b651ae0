"""Dependency-free reader (and minimal writer) for the HDF5 files of the reference's data path.

The reference stores images and tag vectors in HDF5 through h5py (`utils/dataset.py:332-346`:
`h.attrs['captions_per_image']`, `h.create_dataset('images', (N, 3, 256, 256), dtype='uint8')`,
`t.attrs['tag_size']`, `t.create_dataset('tags', (N, tag_size), dtype='float32')`) and reads them back
row by row (`datasets/caption.py:26-29,51`, `datasets/tag.py:24-34`).  h5py/libhdf5 are not part of this
stack, and the access pattern here is different anyway: the train loop wants whole uint8 batches gathered
straight out of the page cache into a pinned staging buffer (`scnattn.data`), which a memory map of the
dataset's contiguous storage gives for free.

This module therefore parses just enough of the HDF5 file format (spec v1.10/3.0) to locate root-group
datasets and attributes:
  * superblock versions 0-3, 8-byte or 4-byte offsets/lengths;
  * version-1 object headers with continuation blocks and "old style" groups (symbol-table message,
    version-1 group B-tree, SNOD nodes, local heap) — what h5py writes by default — and version-2
    object headers ("OHDR"/"OCHK") with compact link messages (libver='latest');
  * dataspace v1/v2, fixed-point and IEEE floating-point datatypes, attribute messages v1-v3;
  * data layout v3/v4 contiguous and compact, and v3 chunked storage without filters (version-1 chunk
    B-tree).  Filtered (compressed) chunks, v4 chunk indices, dense groups, variable-length and compound
    types are refused with an explanatory error: the reference never writes them.
Checked against files written by real h5py 3.3 / HDF5 1.10.6 (tests/golden/hdf5, oracle/gen_hdf5_golden.py).
"""
import mmap
import os
import struct

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"


class H5Error(IOError):
    pass


def _undef(v, size):
    return v == (1 << (8 * size)) - 1


class _Reader:
    def __init__(self, buf):
        self.b = buf
        self.O = 8
        self.L = 8
        self.base = 0

    def u(self, off, n):
        if off < 0 or off > len(self.b) or off + n > len(self.b):
            raise H5Error("truncated HDF5 file (read of %d bytes at offset %d past end %d)" % (n, off, len(self.b)))
        return int.from_bytes(self.b[off:off + n], "little")

    def off(self, pos):
        return self.u(pos, self.O)

    def len_(self, pos):
        return self.u(pos, self.L)

    def bytes(self, off, n):
        if off < 0 or off + n > len(self.b):
            raise H5Error("truncated HDF5 file")
        return bytes(self.b[off:off + n])

    def cstr(self, off, limit=4096):
        if off < 0 or off >= len(self.b):
            raise H5Error("string offset outside the file (corrupt HDF5 structure)")
        end = self.b.find(b"\0", off, min(off + limit, len(self.b)))
        if end < 0:
            raise H5Error("unterminated string in HDF5 heap")
        try:
            return bytes(self.b[off:end]).decode("utf-8")
        except UnicodeDecodeError:
            raise H5Error("name is not valid UTF-8 (corrupt HDF5 structure)")

    def budget(self, what="structure"):
        """Every object-header block / B-tree node visited draws on one budget, so that cyclic or absurdly
        wide (corrupt) structures end in an error instead of a hang."""
        self.visits = getattr(self, "visits", 0) + 1
        if self.visits > 200000:
            raise H5Error("HDF5 %s too large or cyclic (corrupt file?)" % what)


def _pad8(n):
    return (n + 7) & ~7


def _parse_datatype(r, pos):
    """-> (numpy dtype, encoded size).  Datatype message (type 0x03)."""
    cv = r.u(pos, 1)
    cls, ver = cv & 0x0F, cv >> 4
    bits0 = r.u(pos + 1, 1)
    size = r.u(pos + 4, 4)
    if ver not in (1, 2, 3):
        raise H5Error("unsupported datatype message version %d" % ver)
    order = ">" if (bits0 & 1) else "<"
    if cls == 0:   # fixed point; properties: bit offset (2), precision (2)
        if size not in (1, 2, 4, 8):
            raise H5Error("unsupported integer size %d" % size)
        kind = "i" if (bits0 & 0x08) else "u"
        return np.dtype("%s%s%d" % (order if size > 1 else "|", kind, size)), 8 + 4
    if cls == 1:   # floating point; properties 12 bytes
        if size not in (2, 4, 8):
            raise H5Error("unsupported float size %d" % size)
        return np.dtype("%sf%d" % (order, size)), 8 + 12
    names = {2: "time", 3: "string", 4: "bitfield", 5: "opaque", 6: "compound", 7: "reference", 8: "enum",
             9: "variable-length", 10: "array"}
    raise H5Error("unsupported HDF5 datatype class %d (%s): only integer and IEEE float data are read"
                  % (cls, names.get(cls, "?")))


def _parse_dataspace(r, pos):
    """-> shape tuple.  Dataspace message (type 0x01), versions 1 and 2."""
    ver = r.u(pos, 1)
    rank = r.u(pos + 1, 1)
    flags = r.u(pos + 2, 1)
    if ver == 1:
        p = pos + 8
    elif ver == 2:
        if r.u(pos + 3, 1) == 2:
            raise H5Error("null dataspace")
        p = pos + 4
    else:
        raise H5Error("unsupported dataspace message version %d" % ver)
    if rank > 32:
        raise H5Error("dataspace of rank %d (corrupt HDF5 structure)" % rank)
    shape = tuple(r.len_(p + i * r.L) for i in range(rank))
    total = 1
    for dim in shape:
        total *= dim
        if total > (1 << 48):
            raise H5Error("dataspace with an implausible element count (corrupt HDF5 structure)")
    size = (p - pos) + rank * r.L * (2 if flags & 1 else 1)
    return shape, size


class _Object:
    """Messages of one object header, as (type, data offset, size, flags)."""

    def __init__(self, r, addr):
        self.r = r
        self.addr = addr
        self.msgs = []
        b = r.b
        if b[addr:addr + 4] == b"OHDR":
            self._v2(addr)
        else:
            self._v1(addr)

    def _v1(self, addr):
        r = self.r
        if r.u(addr, 1) != 1:
            raise H5Error("unsupported object header version %d at %d" % (r.u(addr, 1), addr))
        nmsg = r.u(addr + 2, 2)
        hsize = r.u(addr + 8, 4)
        blocks = [(addr + 16, hsize)]   # 12 bytes of prefix + 4 of alignment padding
        seen = 0
        while blocks and seen < nmsg:
            r.budget("object header")
            pos, left = blocks.pop(0)
            end = pos + left
            while pos + 8 <= end and seen < nmsg:
                mtype, msize, mflags = r.u(pos, 2), r.u(pos + 2, 2), r.u(pos + 4, 1)
                data = pos + 8
                seen += 1
                if mtype == 0x10:    # continuation
                    blocks.append((r.base + r.off(data), r.len_(data + r.O)))
                elif mtype != 0:
                    self.msgs.append((mtype, data, msize, mflags))
                pos = data + msize

    def _v2(self, addr):
        r = self.r
        if r.u(addr + 4, 1) != 2:
            raise H5Error("unsupported object header version")
        flags = r.u(addr + 5, 1)
        pos = addr + 6
        if flags & 0x20:
            pos += 16
        if flags & 0x10:
            pos += 4
        csz = 1 << (flags & 3)
        chunk0 = r.u(pos, csz)
        pos += csz
        track = bool(flags & 0x04)
        blocks = [(pos, chunk0)]
        while blocks:
            r.budget("object header")
            pos, left = blocks.pop(0)
            end = pos + left
            while pos + 4 <= end:
                mtype, msize, mflags = r.u(pos, 1), r.u(pos + 1, 2), r.u(pos + 3, 1)
                data = pos + 4 + (2 if track else 0)
                if data + msize > end:
                    break
                r.budget("object header")
                if mtype == 0x10:
                    caddr, clen = r.base + r.off(data), r.len_(data + r.O)
                    if clen < 8 or r.bytes(caddr, 4) != b"OCHK":
                        raise H5Error("bad object header continuation signature")
                    blocks.append((caddr + 4, clen - 8))   # minus signature and checksum
                elif mtype != 0:
                    self.msgs.append((mtype, data, msize, mflags))
                pos = data + msize

    def find(self, mtype):
        return [m for m in self.msgs if m[0] == mtype]

    # ---- attributes ------------------------------------------------------------------------------
    def attrs(self):
        r = self.r
        out = {}
        for _, pos, size, mflags in self.find(0x0C):
            if mflags & 0x02:
                raise H5Error("shared attribute messages are not supported")
            ver = r.u(pos, 1)
            nsz, tsz, ssz = r.u(pos + 2, 2), r.u(pos + 4, 2), r.u(pos + 6, 2)
            if ver == 1:
                p = pos + 8
                name = r.cstr(p, nsz + 1)
                p += _pad8(nsz)
                tpos, p = p, p + _pad8(tsz)
                spos, p = p, p + _pad8(ssz)
            elif ver in (2, 3):
                if r.u(pos + 1, 1) & 3:
                    raise H5Error("shared datatype/dataspace in attribute is not supported")
                p = pos + 8 + (1 if ver == 3 else 0)
                name = r.cstr(p, nsz + 1)
                p += nsz
                tpos, p = p, p + tsz
                spos, p = p, p + ssz
            else:
                raise H5Error("unsupported attribute message version %d" % ver)
            try:
                dt, _ = _parse_datatype(r, tpos)
            except H5Error:
                continue                     # e.g. string attributes: not needed on this path
            shape, _ = _parse_dataspace(r, spos)
            n = int(np.prod(shape)) if shape else 1
            val = np.frombuffer(r.bytes(p, n * dt.itemsize), dtype=dt).reshape(shape)
            out[name] = val[()] if shape == () else val.copy()
        if self.find(0x15):
            pos = self.find(0x15)[0][1]
            fl = r.u(pos + 1, 1)
            q = pos + 2 + (2 if fl & 1 else 0)
            if not _undef(r.off(q), r.O):
                raise H5Error("attributes in dense storage (fractal heap) are not supported")
        return out

    # ---- group links -----------------------------------------------------------------------------
    def links(self):
        r = self.r
        out = {}
        st = self.find(0x11)
        if st:                               # old-style group: B-tree of symbol nodes + local heap
            btree = r.base + r.off(st[0][1])
            heap = r.base + r.off(st[0][1] + r.O)
            if r.bytes(heap, 4) != b"HEAP":
                raise H5Error("bad local heap signature")
            hdata = r.base + r.off(heap + 8 + 2 * r.L)
            self._walk_group_btree(btree, hdata, out, 0)
            return out
        for _, pos, size, _f in self.find(0x06):   # compact new-style group: link messages
            if r.u(pos, 1) != 1:
                raise H5Error("unsupported link message version")
            fl = r.u(pos + 1, 1)
            p = pos + 2
            ltype = 0
            if fl & 0x08:
                ltype = r.u(p, 1)
                p += 1
            if fl & 0x04:
                p += 8
            if fl & 0x10:
                p += 1
            lsz = 1 << (fl & 3)
            nlen = r.u(p, lsz)
            p += lsz
            name = r.bytes(p, nlen).decode("utf-8")
            p += nlen
            if ltype == 0:
                out[name] = r.base + r.off(p)
        li = self.find(0x02)
        if li and not out:
            pos = li[0][1]
            fl = r.u(pos + 1, 1)
            q = pos + 2 + (8 if fl & 1 else 0)
            if not _undef(r.off(q), r.O):
                raise H5Error("group with dense link storage (fractal heap) is not supported; the reference "
                              "writes one or two datasets per file")
        return out

    def _walk_group_btree(self, addr, hdata, out, depth):
        r = self.r
        r.budget("group B-tree")
        if depth > 16:
            raise H5Error("group B-tree too deep")
        if r.bytes(addr, 4) != b"TREE" or r.u(addr + 4, 1) != 0:
            raise H5Error("bad group B-tree node at %d" % addr)
        level, used = r.u(addr + 5, 1), r.u(addr + 6, 2)
        p = addr + 8 + 2 * r.O
        for i in range(used):
            child = r.base + r.off(p + r.L + i * (r.L + r.O))
            if level > 0:
                self._walk_group_btree(child, hdata, out, depth + 1)
                continue
            if r.bytes(child, 4) != b"SNOD":
                raise H5Error("bad symbol table node at %d" % child)
            n = r.u(child + 6, 2)
            esz = 2 * r.O + 8 + 16
            for k in range(n):
                e = child + 8 + k * esz
                out[r.cstr(hdata + r.off(e))] = r.base + r.off(e + r.O)


_PARSE_ERRORS = (ValueError, OverflowError, IndexError, struct.error, MemoryError, RecursionError, TypeError)


def _guard(fn):
    """Anything a corrupt byte can provoke inside the parser surfaces as H5Error."""
    def wrapped(*a, **k):
        try:
            return fn(*a, **k)
        except H5Error:
            raise
        except KeyError:
            raise
        except _PARSE_ERRORS as e:
            raise H5Error("corrupt or unsupported HDF5 structure (%s: %s)" % (type(e).__name__, e))
    wrapped.__name__ = fn.__name__
    wrapped.__doc__ = fn.__doc__
    return wrapped


class Dataset:
    """Read-only view of one dataset.  `ds[i]`, `ds[a:b]`, `ds[[3, 1, 2]]` return numpy arrays like h5py;
    `ds.array` is the zero-copy memory map (contiguous layout) used by the batch loader."""

    @_guard
    def __init__(self, f, name, obj):
        r = f._r
        self.file, self.name, self._obj = f, name, obj
        sp, dt, lay = obj.find(0x01), obj.find(0x03), obj.find(0x08)
        if not (sp and dt and lay):
            raise H5Error("'%s' is not a dataset" % name)
        if dt[0][3] & 0x02:
            raise H5Error("committed (shared) datatypes are not supported")
        self.shape, _ = _parse_dataspace(r, sp[0][1])
        self.dtype, _ = _parse_datatype(r, dt[0][1])
        self.attrs = obj.attrs()
        self.layout = None
        self._array = None
        self._chunks = None
        pos = lay[0][1]
        ver, cls = r.u(pos, 1), r.u(pos + 1, 1)
        if ver not in (3, 4):
            raise H5Error("unsupported data layout message version %d" % ver)
        nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize if self.shape else self.dtype.itemsize
        if cls == 1:
            self.layout = "contiguous"
            addr = r.off(pos + 2)
            if _undef(addr, r.O):            # never written (late allocation): fill value, zeros by default
                self._array = np.zeros(self.shape, self.dtype)
            else:
                if r.len_(pos + 2 + r.O) < nbytes or r.base + addr + nbytes > len(r.b):
                    raise H5Error("dataset '%s' extends past the end of the file (truncated?)" % name)
                self._array = np.frombuffer(r.b, dtype=self.dtype, count=nbytes // self.dtype.itemsize,
                                            offset=r.base + addr).reshape(self.shape)
        elif cls == 0:
            self.layout = "compact"
            n = r.u(pos + 2, 2)
            self._array = np.frombuffer(r.bytes(pos + 4, n), dtype=self.dtype).reshape(self.shape)
        elif cls == 2:
            self.layout = "chunked"
            if obj.find(0x0B):
                raise H5Error("dataset '%s' is stored in filtered (compressed) chunks; only uncompressed data is "
                              "read — the reference's dataset builder writes contiguous uint8/float32 arrays "
                              "(utils/dataset.py:340-346)" % name)
            if ver != 3:
                raise H5Error("dataset '%s' uses a version-4 chunk index (libver='latest'); rewrite it "
                              "contiguous or with default h5py settings" % name)
            ndim = r.u(pos + 2, 1)
            btree = r.off(pos + 3)
            cdims = tuple(r.u(pos + 3 + r.O + 4 * i, 4) for i in range(ndim))
            if ndim != len(self.shape) + 1 or cdims[-1] != self.dtype.itemsize:
                raise H5Error("inconsistent chunk dimensions")
            self._chunks = (None if _undef(btree, r.O) else r.base + btree, cdims[:-1])
        else:
            raise H5Error("unsupported data layout class %d" % cls)

    # ---- chunked storage, unfiltered -------------------------------------------------------------------
    @_guard
    def _load_chunked(self):
        r = self.file._r
        btree, cdims = self._chunks
        nbytes = int(np.prod(self.shape, dtype=np.float64)) * self.dtype.itemsize
        if nbytes > 64 * len(r.b) + (1 << 20) or any(c <= 0 for c in cdims):
            raise H5Error("chunked dataset '%s' claims %d bytes in a %d-byte file (corrupt?)" % (self.name, nbytes, len(r.b)))
        out = np.zeros(self.shape, self.dtype)
        if btree is None:
            return out
        rank = len(self.shape)

        def walk(addr, depth):
            r.budget("chunk B-tree")
            if depth > 32 or r.bytes(addr, 4) != b"TREE" or r.u(addr + 4, 1) != 1:
                raise H5Error("bad chunk B-tree node at %d" % addr)
            level, used = r.u(addr + 5, 1), r.u(addr + 6, 2)
            ksz = 8 + 8 * (rank + 1)
            p = addr + 8 + 2 * r.O
            for i in range(used):
                k = p + i * (ksz + r.O)
                csize, mask = r.u(k, 4), r.u(k + 4, 4)
                offs = tuple(r.u(k + 8 + 8 * j, 8) for j in range(rank))
                child = r.base + r.off(k + ksz)
                if level > 0:
                    walk(child, depth + 1)
                    continue
                if mask:
                    raise H5Error("chunk with a filter mask")
                n = int(np.prod(cdims)) * self.dtype.itemsize
                if csize != n:
                    raise H5Error("chunk size mismatch (filtered data?)")
                blk = np.frombuffer(r.bytes(child, n), dtype=self.dtype).reshape(cdims)
                sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, self.shape))
                out[sl] = blk[tuple(slice(0, s.stop - s.start) for s in sl)]
        walk(btree, 0)
        return out

    @property
    def array(self):
        if self._array is None:
            self._array = self._load_chunked()
            self._array.setflags(write=False)
        return self._array

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, idx):
        a = self.array[idx]
        return np.array(a) if isinstance(a, np.ndarray) else a     # a copy, like h5py

    def __repr__(self):
        return '<h5lite dataset "%s": shape %s, type "%s", %s>' % (self.name, self.shape, self.dtype.str, self.layout)


class File:
    """`File(path)` opens read-only; `f['images']`, `f.attrs['captions_per_image']`, `f.keys()`,
    context manager — the subset of the h5py API the reference uses (datasets/caption.py:26-32)."""

    def __init__(self, path, mode="r"):
        if mode != "r":
            raise ValueError("h5lite.File is read-only; use h5lite.write_arrays to create files")
        self.filename = path
        self._fh = open(path, "rb")
        size = os.fstat(self._fh.fileno()).st_size
        if size < 32:
            self._fh.close()
            raise H5Error("%s: not an HDF5 file (too short)" % path)
        self._mm = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        try:
            self._open()
        except Exception:
            self.close()
            raise

    @_guard
    def _open(self):
        mm = self._mm
        start = 0
        while mm[start:start + 8] != _SIG:     # the superblock may sit at 0, 512, 1024, ... (user block)
            start = 512 if start == 0 else start * 2
            if start + 8 > len(mm):
                raise H5Error("%s: HDF5 signature not found" % self.filename)
        r = self._r = _Reader(mm)
        ver = r.u(start + 8, 1)
        if ver in (0, 1):
            r.O, r.L = r.u(start + 13, 1), r.u(start + 14, 1)
            p = start + 24 + (4 if ver == 1 else 0)
            base = r.off(p)
            ste = p + 4 * r.O                  # root group symbol table entry
            root = r.off(ste + r.O)
        elif ver in (2, 3):
            r.O, r.L = r.u(start + 9, 1), r.u(start + 10, 1)
            base = r.off(start + 12)
            root = r.off(start + 12 + 3 * r.O)
        else:
            raise H5Error("unsupported HDF5 superblock version %d" % ver)
        if r.O not in (4, 8) or r.L not in (4, 8):
            raise H5Error("unsupported offset/length size")
        r.base = base if ver >= 2 or base else start
        if _undef(root, r.O):
            raise H5Error("file has no root group")
        self._root = _Object(r, r.base + root)
        self._links = self._root.links()
        self.attrs = self._root.attrs()
        self._cache = {}

    def keys(self):
        return sorted(self._links)

    def __contains__(self, name):
        return name in self._links

    @_guard
    def __getitem__(self, name):
        name = name.lstrip("/")
        if name not in self._links:
            raise KeyError("Unable to open object (object '%s' doesn't exist)" % name)
        if name not in self._cache:
            self._cache[name] = Dataset(self, name, _Object(self._r, self._links[name]))
        return self._cache[name]

    def close(self):
        # arrays handed out by Dataset.array keep the map alive (numpy holds a buffer export); closing is
        # then deferred to garbage collection, as with any np.memmap
        self._cache = {}
        try:
            self._mm.close()
        except (BufferError, ValueError):
            pass
        self._fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


# ---- minimal writer ---------------------------------------------------------------------------------
def write_arrays(path, datasets, attrs=None):
    """Write root-level datasets (name -> numpy integer/float array, stored contiguous) and scalar integer
    root attributes in the layout h5py's defaults produce (superblock 0, version-1 object headers, one
    symbol-table node), so that h5py / libhdf5 read the result.  Enough to build the reference's
    `*_IMAGES_*.hdf5` / `*_TAGS_*.hdf5` without h5py; at most 8 datasets (one SNOD leaf)."""
    attrs = dict(attrs or {})
    names = sorted(datasets)
    if not 1 <= len(names) <= 8:
        raise ValueError("write_arrays stores 1..8 datasets")
    arrs = {k: np.ascontiguousarray(datasets[k]) for k in names}

    def dt_msg(dt):
        dt = np.dtype(dt)
        if dt.kind in "iu":
            bits = 0x08 if dt.kind == "i" else 0
            return struct.pack("<BBBBIHH", 0x10, bits, 0, 0, dt.itemsize, 0, 8 * dt.itemsize)
        if dt.kind == "f" and dt.itemsize in (4, 8):
            if dt.itemsize == 4:
                return struct.pack("<BBBBIHHBBBBI", 0x11, 0x20, 31, 0, 4, 0, 32, 23, 8, 0, 23, 127)
            return struct.pack("<BBBBIHHBBBBI", 0x11, 0x20, 63, 0, 8, 0, 64, 52, 11, 0, 52, 1023)
        raise ValueError("write_arrays: unsupported dtype %s" % dt)

    def msg(mtype, body, flags=0):
        body = body + b"\0" * (_pad8(len(body)) - len(body))
        return struct.pack("<HHB3x", mtype, len(body), flags) + body

    def header(msgs):
        body = b"".join(msgs)
        return struct.pack("<BxHII4x", 1, len(msgs), 1, len(body)) + body

    # local heap data: offset 0 = "" (root), then names
    heap_data = bytearray(b"\0" * 8)
    name_off = {}
    for k in names:
        name_off[k] = len(heap_data)
        nb = k.encode("utf-8") + b"\0"
        heap_data += nb + b"\0" * (_pad8(len(nb)) - len(nb))
    heap_size = _pad8(len(heap_data)) + 16
    heap_data += b"\0" * (heap_size - len(heap_data))
    free_off = heap_size - 16
    struct.pack_into("<QQ", heap_data, free_off, 1, 16)      # free block: next = 1 (last), size

    def attr_msg(name, value):
        v = np.asarray(value)
        if v.shape != () or v.dtype.kind not in "iu":
            raise ValueError("write_arrays: only scalar integer attributes")
        v = v.astype("<i8")
        nb = name.encode("utf-8") + b"\0"
        dtb = dt_msg(v.dtype)
        body = struct.pack("<BxHHH", 1, len(nb), len(dtb), 8)
        body += nb + b"\0" * (_pad8(len(nb)) - len(nb))
        body += dtb + b"\0" * (_pad8(len(dtb)) - len(dtb))
        body += struct.pack("<BBB5x", 1, 0, 0)
        body += v.tobytes()
        return msg(0x0C, body)

    # layout: superblock (96) | root header | B-tree node | heap header+data | SNOD | dataset headers | data
    SB = 96
    root_msgs_len = 8 + 16 + sum(len(attr_msg(k, v)) for k, v in attrs.items())
    root_addr = SB
    btree_addr = _pad8(root_addr + 16 + root_msgs_len)
    K_LEAF, K_INT = 4, 16
    btree_size = 8 + 16 + (2 * K_INT) * 16 + 8
    heap_addr = btree_addr + btree_size
    heap_data_addr = heap_addr + 32
    snod_addr = heap_data_addr + heap_size
    snod_size = 8 + (2 * K_LEAF) * 40
    pos = snod_addr + snod_size
    ds_addr, ds_hdr = {}, {}
    for k in names:
        ds_addr[k] = pos
        a = arrs[k]
        sp = struct.pack("<BBB5x", 1, a.ndim, 1) + b"".join(struct.pack("<Q", s) for s in a.shape) * 2
        ds_hdr[k] = [msg(0x01, sp), msg(0x03, dt_msg(a.dtype), flags=1), msg(0x05, struct.pack("<BBBB", 2, 2, 2, 0)),
                     None]
        pos += 16 + len(ds_hdr[k][0]) + len(ds_hdr[k][1]) + len(ds_hdr[k][2]) + 8 + 24
    data_addr = {}
    pos = (pos + 2047) & ~2047 if any(a.nbytes >= 2048 for a in arrs.values()) else pos
    for k in names:
        data_addr[k] = pos if arrs[k].nbytes else (1 << 64) - 1
        pos += _pad8(arrs[k].nbytes)
    eof = pos

    with open(path, "wb") as fh:
        sb = _SIG + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, K_LEAF, K_INT, 0)
        sb += struct.pack("<QQQQ", 0, (1 << 64) - 1, eof, (1 << 64) - 1)
        sb += struct.pack("<QQII", 0, root_addr, 1, 0) + struct.pack("<QQ", btree_addr, heap_addr)
        assert len(sb) == SB
        fh.write(sb)
        root = header([msg(0x11, struct.pack("<QQ", btree_addr, heap_addr))] + [attr_msg(k, v) for k, v in attrs.items()])
        fh.write(root)
        fh.write(b"\0" * (btree_addr - fh.tell()))
        bt = b"TREE" + struct.pack("<BBH", 0, 0, 1) + struct.pack("<QQ", (1 << 64) - 1, (1 << 64) - 1)
        bt += struct.pack("<QQQ", 0, snod_addr, name_off[names[-1]])
        fh.write(bt + b"\0" * (btree_size - len(bt)))
        fh.write(b"HEAP" + struct.pack("<B3xQQQ", 0, heap_size, free_off, heap_data_addr))
        fh.write(bytes(heap_data))
        sn = b"SNOD" + struct.pack("<BxH", 1, len(names))
        for k in names:
            sn += struct.pack("<QQII16x", name_off[k], ds_addr[k], 0, 0)
        fh.write(sn + b"\0" * (snod_size - len(sn)))
        for k in names:
            assert fh.tell() == ds_addr[k]
            lay = msg(0x08, struct.pack("<BBQQ", 3, 1, data_addr[k], arrs[k].nbytes))
            fh.write(header(ds_hdr[k][:3] + [lay]))
        for k in names:
            if arrs[k].nbytes:
                fh.write(b"\0" * (data_addr[k] - fh.tell()))
                fh.write(arrs[k].tobytes() if arrs[k].nbytes < (1 << 26) else memoryview(arrs[k]).cast("B"))
        fh.write(b"\0" * (eof - fh.tell()))
