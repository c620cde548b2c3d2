"""
A read-only HDF5 reader in pure Python + numpy + zlib, sufficient for GANce projection files.

The reference reads projections with h5py (gance/projection/projection_file_reader.py:102-152), which is not
installed where this package runs. The files its writer produces (gance/projection/projector_file_writer.py:726-834,
h5py defaults: `libver="earliest"`) use a small, fixed subset of the HDF5 file format, restated here from the
published HDF5 File Format Specification (version 1.1 structures):

  * superblock version 0 / 1, 8-byte offsets and lengths;
  * version-1 object headers (+ continuation blocks);
  * old-style groups: symbol-table message -> version-1 B-tree (node type 0) + local heap + symbol-table nodes;
  * datasets: dataspace message v1 / v2, datatype classes fixed-point / floating-point / string / enum /
    variable-length string / array, data layout message v3 (compact, contiguous, chunked through a version-1
    B-tree of node type 1), filter pipeline v1 / v2 with shuffle (2) and deflate (1);
  * attributes (message versions 1-3) of those datatypes, variable-length strings through global heap collections.

Anything outside that subset (new-style groups, version-2 object headers, other filters, compound types ...)
raises `UnsupportedHdf5` with the structure's name: such a file was not written by the reference's writer with
h5py's defaults. Pinned by REAL files written with h5py (oracle/make_hdf5_fixture.py, tests/golden/*.hdf5).

API (the part of h5py's the projection reader uses): `File(path)` -> `.attrs` (dict), `file[name]`, `.keys()`,
`group[name]`, `np.array(dataset)`, `.close()`.
"""

import struct
import zlib
from pathlib import Path
from typing import Any, Dict, Iterator, List, Optional, Tuple

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEFINED = 0xFFFFFFFFFFFFFFFF


class UnsupportedHdf5(RuntimeError):
    """The file uses an HDF5 structure the reference's writer never produces."""


class _Datatype:
    """A parsed datatype message."""

    def __init__(self, kind: str, size: int, dtype: Optional[np.dtype] = None, base: Optional["_Datatype"] = None,
                 dims: Tuple[int, ...] = (), enum: Optional[Dict[int, str]] = None, charset: int = 0) -> None:
        self.kind, self.size, self.dtype, self.base, self.dims, self.enum, self.charset = kind, size, dtype, base, dims, enum, charset


class _Reader:
    """The file image and the low-level structure parsers."""

    def __init__(self, data: bytes) -> None:
        self.data = data
        if data[:8] != SIGNATURE:
            raise UnsupportedHdf5("not an HDF5 file (superblock signature at offset 0)")
        version = data[8]
        if version not in (0, 1):
            raise UnsupportedHdf5(f"superblock version {version} (the reference's writer uses h5py defaults: version 0)")
        if data[13] != 8 or data[14] != 8:
            raise UnsupportedHdf5("offsets / lengths that are not 8 bytes wide")
        pos = 24 + (4 if version == 1 else 0)
        self.base_address = self.u64(pos)
        # root group symbol table entry follows base, free-space, end-of-file and driver addresses
        self.root_header = self.u64(pos + 32 + 8)
        self._heap_cache: Dict[int, Dict[int, bytes]] = {}

    def u8(self, pos: int) -> int:
        return self.data[pos]

    def u16(self, pos: int) -> int:
        return struct.unpack_from("<H", self.data, pos)[0]

    def u32(self, pos: int) -> int:
        return struct.unpack_from("<I", self.data, pos)[0]

    def u64(self, pos: int) -> int:
        return struct.unpack_from("<Q", self.data, pos)[0]

    # ---- object headers ----

    def messages(self, address: int) -> List[Tuple[int, int, int]]:
        """(type, offset of the message data, size) of every message of a version-1 object header."""
        if self.data[address : address + 4] == b"OHDR":
            raise UnsupportedHdf5("version-2 object header (file written with libver='latest')")
        if self.u8(address) != 1:
            raise UnsupportedHdf5(f"object header version {self.u8(address)}")
        count = self.u16(address + 2)
        size = self.u32(address + 8)
        blocks = [(address + 16, size)]
        out: List[Tuple[int, int, int]] = []
        while blocks and len(out) < count:
            pos, remaining = blocks.pop(0)
            end = pos + remaining
            while pos + 8 <= end and len(out) < count:
                mtype, msize = self.u16(pos), self.u16(pos + 2)
                body = pos + 8
                out.append((mtype, body, msize))
                if mtype == 0x0010:  # continuation
                    blocks.append((self.u64(body), self.u64(body + 8)))
                pos = body + msize
        return out

    # ---- groups ----

    def local_heap_string(self, heap_address: int, offset: int) -> str:
        if self.data[heap_address : heap_address + 4] != b"HEAP":
            raise UnsupportedHdf5("local heap signature")
        segment = self.u64(heap_address + 24)
        start = segment + offset
        end = self.data.index(b"\x00", start)
        return self.data[start:end].decode("utf-8")

    def group_links(self, btree: int, heap: int) -> Dict[str, int]:
        """name -> object header address, walking the version-1 B-tree of a symbol-table group."""
        links: Dict[str, int] = {}
        if btree == UNDEFINED:
            return links
        if self.data[btree : btree + 4] != b"TREE" or self.u8(btree + 4) != 0:
            raise UnsupportedHdf5("group B-tree node")
        level, used = self.u8(btree + 5), self.u16(btree + 6)
        pos = btree + 24
        for index in range(used):
            child = self.u64(pos + 8 + index * 16)
            if level > 0:
                links.update(self.group_links(child, heap))
                continue
            if self.data[child : child + 4] != b"SNOD":
                raise UnsupportedHdf5("symbol table node signature")
            for entry in range(self.u16(child + 6)):
                at = child + 8 + entry * 40
                links[self.local_heap_string(heap, self.u64(at))] = self.u64(at + 8)
        return links

    # ---- datatypes / dataspaces ----

    def datatype(self, pos: int) -> Tuple[_Datatype, int]:
        """Parse the datatype message at `pos`; returns it and the number of bytes consumed."""
        class_version = self.u8(pos)
        cls, version = class_version & 0x0F, class_version >> 4
        bits = self.u8(pos + 1) | (self.u8(pos + 2) << 8) | (self.u8(pos + 3) << 16)
        size = self.u32(pos + 4)
        body = pos + 8
        if cls == 0:  # fixed point
            if bits & 1:
                raise UnsupportedHdf5("big-endian integers")
            return _Datatype("int", size, np.dtype(("<i" if bits & 8 else "<u") + str(size))), 8 + 4
        if cls == 1:  # floating point
            if bits & 1:
                raise UnsupportedHdf5("big-endian floats")
            return _Datatype("float", size, np.dtype("<f" + str(size))), 8 + 12
        if cls == 3:  # fixed-length string
            return _Datatype("string", size, np.dtype("S" + str(size)), charset=(bits >> 4) & 0xF), 8
        if cls == 8:  # enumeration (h5py's booleans: FALSE = 0, TRUE = 1 over int8)
            members = bits & 0xFFFF
            base, used = self.datatype(body)
            at = body + used
            names = []
            for _ in range(members):
                end = self.data.index(b"\x00", at)
                names.append(self.data[at:end].decode("utf-8"))
                length = end - at + 1
                at += length if version >= 3 else (length + 7) // 8 * 8
            values = np.frombuffer(self.data, dtype=base.dtype, count=members, offset=at)
            at += members * base.size
            return _Datatype("enum", size, base.dtype, base=base, enum={int(v): n for v, n in zip(values, names)}), at - pos
        if cls == 9:  # variable length
            if (bits & 0xF) != 1:
                raise UnsupportedHdf5("variable-length sequences (only variable-length strings are read)")
            base, used = self.datatype(body)
            return _Datatype("vlen_string", size, base=base, charset=(bits >> 8) & 0xF), 8 + used
        if cls == 10:  # array
            rank = self.u8(body)
            at = body + (4 if version < 3 else 1)
            dims = tuple(self.u32(at + 4 * i) for i in range(rank))
            at += 4 * rank * (2 if version < 3 else 1)  # version 2 carries permutation indices as well
            base, used = self.datatype(at)
            return _Datatype("array", size, base=base, dims=dims), at + used - pos
        raise UnsupportedHdf5(f"datatype class {cls}")

    def dataspace(self, pos: int) -> Tuple[int, ...]:
        version = self.u8(pos)
        rank, flags = self.u8(pos + 1), self.u8(pos + 2)
        if version == 1:
            at = pos + 8
        elif version == 2:
            if self.u8(pos + 3) == 2:
                raise UnsupportedHdf5("null dataspace")
            at = pos + 4
        else:
            raise UnsupportedHdf5(f"dataspace version {version}")
        del flags
        return tuple(self.u64(at + 8 * i) for i in range(rank))

    def global_heap_object(self, collection: int, index: int) -> bytes:
        if collection not in self._heap_cache:
            if self.data[collection : collection + 4] != b"GCOL":
                raise UnsupportedHdf5("global heap collection signature")
            size = self.u64(collection + 8)
            objects: Dict[int, bytes] = {}
            at, end = collection + 16, collection + size
            while at + 16 <= end:
                obj_index, obj_size = self.u16(at), self.u64(at + 8)
                if obj_index == 0:
                    break
                objects[obj_index] = self.data[at + 16 : at + 16 + obj_size]
                at += 16 + (obj_size + 7) // 8 * 8
            self._heap_cache[collection] = objects
        return self._heap_cache[collection][index]

    def decode(self, dtype: _Datatype, shape: Tuple[int, ...], raw: bytes) -> Any:
        """Raw element bytes -> numpy array / Python value of `shape` (() = scalar)."""
        count = int(np.prod(shape)) if shape else 1
        if dtype.kind in ("int", "float"):
            array = np.frombuffer(raw, dtype=dtype.dtype, count=count).reshape(shape)
            return array[()] if not shape else array.copy()
        if dtype.kind == "enum":
            array = np.frombuffer(raw, dtype=dtype.dtype, count=count).reshape(shape)
            if dtype.enum is not None and set(dtype.enum.values()) == {"FALSE", "TRUE"}:
                array = array.astype(bool)
            return array[()] if not shape else array.copy()
        if dtype.kind == "string":
            items = [raw[i * dtype.size : (i + 1) * dtype.size].split(b"\x00")[0] for i in range(count)]
            values = [item.decode("utf-8") if dtype.charset == 1 else item for item in items]
            return values[0] if not shape else np.array(values, dtype=object).reshape(shape)
        if dtype.kind == "vlen_string":
            values = []
            for i in range(count):
                length, collection, index = struct.unpack_from("<IQI", raw, i * 16)
                text = self.global_heap_object(collection, index)[:length] if collection not in (0, UNDEFINED) else b""
                values.append(text.decode("utf-8"))
            return values[0] if not shape else np.array(values, dtype=object).reshape(shape)
        if dtype.kind == "array":
            return self.decode(dtype.base, tuple(shape) + dtype.dims, raw)
        raise UnsupportedHdf5(f"values of datatype {dtype.kind}")

    def attribute(self, pos: int) -> Tuple[str, Any]:
        version = self.u8(pos)
        name_size, dt_size, ds_size = self.u16(pos + 2), self.u16(pos + 4), self.u16(pos + 6)
        if version == 1:
            at = pos + 8

            def pad(n: int) -> int:
                return (n + 7) // 8 * 8
        elif version in (2, 3):
            if self.u8(pos + 1) & 0x03:
                raise UnsupportedHdf5("shared attribute datatype / dataspace")
            at = pos + (9 if version == 3 else 8)

            def pad(n: int) -> int:
                return n
        else:
            raise UnsupportedHdf5(f"attribute message version {version}")
        name = self.data[at : at + name_size].split(b"\x00")[0].decode("utf-8")
        at += pad(name_size)
        dtype, _ = self.datatype(at)
        at += pad(dt_size)
        shape = self.dataspace(at)
        at += pad(ds_size)
        count = int(np.prod(shape)) if shape else 1
        return name, self.decode(dtype, shape, self.data[at : at + count * dtype.size])


class Dataset:
    """A dataset; `np.array(dataset)` reads it."""

    def __init__(self, reader: _Reader, name: str, address: int) -> None:
        self._reader, self.name = reader, name
        self._dtype: Optional[_Datatype] = None
        self.shape: Tuple[int, ...] = ()
        self._layout: Optional[Tuple[int, int, int]] = None
        self._filters: List[Tuple[int, List[int]]] = []
        self.attrs: Dict[str, Any] = {}
        for mtype, body, size in reader.messages(address):
            if mtype == 0x0001:
                self.shape = reader.dataspace(body)
            elif mtype == 0x0003:
                self._dtype, _ = reader.datatype(body)
            elif mtype == 0x0008:
                self._layout = (body, size, reader.u8(body))
            elif mtype == 0x000B:
                self._filters = self._parse_filters(body)
            elif mtype == 0x000C:
                key, value = reader.attribute(body)
                self.attrs[key] = value
        if self._dtype is None or self._layout is None:
            raise UnsupportedHdf5(f"{name}: dataset without datatype or layout message")

    @property
    def dtype(self) -> np.dtype:
        return self._dtype.dtype

    def _parse_filters(self, pos: int) -> List[Tuple[int, List[int]]]:
        reader = self._reader
        version, count = reader.u8(pos), reader.u8(pos + 1)
        at = pos + (8 if version == 1 else 2)
        out = []
        for _ in range(count):
            filter_id = reader.u16(at)
            if version == 1 or filter_id >= 256:
                name_length = reader.u16(at + 2)
                at += 2
            else:
                name_length = 0
            values_count = reader.u16(at + 4)
            at += 6
            at += (name_length + 7) // 8 * 8 if version == 1 else name_length
            values = [reader.u32(at + 4 * i) for i in range(values_count)]
            at += 4 * values_count
            if version == 1 and values_count % 2:
                at += 4
            out.append((filter_id, values))
        return out

    def _unfilter(self, chunk: bytes, mask: int) -> bytes:
        for index in reversed(range(len(self._filters))):
            if mask & (1 << index):
                continue
            filter_id, values = self._filters[index]
            if filter_id == 1:
                chunk = zlib.decompress(chunk)
            elif filter_id == 2:
                width = values[0] if values else self._dtype.size
                if width > 1:
                    chunk = np.frombuffer(chunk, dtype=np.uint8).reshape(width, -1).T.tobytes()
            else:
                raise UnsupportedHdf5(f"filter {filter_id} (the reference's writer uses shuffle + gzip only)")
        return chunk

    def _chunks(self, node: int, rank: int) -> Iterator[Tuple[Tuple[int, ...], int, int, int]]:
        """(chunk offset, address, stored size, filter mask) of every chunk under a version-1 chunk B-tree node."""
        reader = self._reader
        if reader.data[node : node + 4] != b"TREE" or reader.u8(node + 4) != 1:
            raise UnsupportedHdf5("chunk B-tree node")
        level, used = reader.u8(node + 5), reader.u16(node + 6)
        key_size = 8 + 8 * (rank + 1)
        pos = node + 24
        for index in range(used):
            key = pos + index * (key_size + 8)
            child = reader.u64(key + key_size)
            if level > 0:
                yield from self._chunks(child, rank)
            else:
                offset = tuple(reader.u64(key + 8 + 8 * d) for d in range(rank))
                yield offset, child, reader.u32(key), reader.u32(key + 4)

    def read(self) -> np.ndarray:
        reader = self._reader
        body, _, version = self._layout
        if version != 3:
            raise UnsupportedHdf5(f"data layout message version {version}")
        layout_class = reader.u8(body + 1)
        dtype = self._dtype
        if dtype.kind not in ("int", "float", "enum"):
            raise UnsupportedHdf5(f"{self.name}: datasets of datatype {dtype.kind}")
        count = int(np.prod(self.shape)) if self.shape else 1
        if layout_class == 0:  # compact
            size = reader.u16(body + 2)
            return reader.decode(dtype, self.shape, reader.data[body + 4 : body + 4 + size])
        if layout_class == 1:  # contiguous
            address = reader.u64(body + 2)
            if address == UNDEFINED:
                return np.zeros(self.shape, dtype=dtype.dtype)
            return reader.decode(dtype, self.shape, reader.data[address : address + count * dtype.size])
        if layout_class != 2:
            raise UnsupportedHdf5(f"data layout class {layout_class}")
        rank = reader.u8(body + 2) - 1
        btree = reader.u64(body + 3)
        chunk_shape = tuple(reader.u32(body + 11 + 4 * d) for d in range(rank))
        out = np.zeros(self.shape, dtype=dtype.dtype)
        if btree == UNDEFINED:
            return out
        for offset, address, stored, mask in self._chunks(btree, rank):
            raw = self._unfilter(reader.data[address : address + stored], mask)
            chunk = np.frombuffer(raw, dtype=dtype.dtype, count=int(np.prod(chunk_shape))).reshape(chunk_shape)
            region = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offset, chunk_shape, self.shape))
            out[region] = chunk[tuple(slice(0, r.stop - r.start) for r in region)]
        return out

    def __array__(self, dtype=None, copy=None) -> np.ndarray:  # pylint: disable=unused-argument
        array = self.read()
        return array.astype(dtype) if dtype is not None else array


class Group:
    """An old-style (symbol table) group."""

    def __init__(self, reader: _Reader, name: str, address: int) -> None:
        self._reader, self.name = reader, name
        self.attrs: Dict[str, Any] = {}
        self._links: Dict[str, int] = {}
        symbol_table = None
        for mtype, body, _ in reader.messages(address):
            if mtype == 0x0011:
                symbol_table = (reader.u64(body), reader.u64(body + 8))
            elif mtype == 0x000C:
                key, value = reader.attribute(body)
                self.attrs[key] = value
            elif mtype in (0x0002, 0x0006):
                raise UnsupportedHdf5("new-style group (link messages): file written with libver='latest'")
        if symbol_table is not None:
            self._links = reader.group_links(*symbol_table)

    def keys(self) -> List[str]:
        return sorted(self._links)

    def __contains__(self, name: str) -> bool:
        return name.strip("/").split("/")[0] in self._links

    def __getitem__(self, name: str):
        head, _, rest = name.strip("/").partition("/")
        address = self._links[head]
        path = f"{self.name.rstrip('/')}/{head}"
        is_group = any(mtype == 0x0011 for mtype, _, _ in self._reader.messages(address))
        item = Group(self._reader, path, address) if is_group else Dataset(self._reader, path, address)
        return item[rest] if rest else item

    def items(self) -> Iterator[Tuple[str, Any]]:
        for key in self.keys():
            yield key, self[key]


class File(Group):
    """Read-only HDF5 file (the whole file is read into memory: projection files without histories are MBs)."""

    def __init__(self, name, mode: str = "r") -> None:
        if mode != "r":
            raise ValueError("hdf5_lite is read-only")
        reader = _Reader(Path(name).read_bytes())
        super().__init__(reader, "/", reader.root_header + reader.base_address)

    def close(self) -> None:
        """Nothing to release (no file handle is kept)."""

    def __enter__(self) -> "File":
        return self

    def __exit__(self, *exc) -> None:
        self.close()
