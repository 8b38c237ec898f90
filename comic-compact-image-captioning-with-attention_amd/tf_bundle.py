"""TensorFlow checkpoint-V2 ("tensor bundle") reader / writer, dependency-free.

The reference saves and restores its models with `tf.train.Saver` (src/train_fn.py:67-70,
:131-132; src/model_base.py:432-482; src/infer_fn.py:103,122), i.e. TF-1.9 tensor bundles:

    <prefix>.index                  an SSTable (LevelDB table format as ported to
                                    tensorflow/core/lib/io/table*.cc): sorted string keys ->
                                    serialized protos; key "" -> BundleHeaderProto, every other
                                    key = variable name -> BundleEntryProto
    <prefix>.data-00000-of-00001    the tensors' raw little-endian bytes, back to back
    checkpoint                      text proto naming the latest prefix (CheckpointState)

[TF-1.9, un-vendored] tensorflow is not installable here and no real checkpoint ships with the
reference, so this container code is written from the published formats and is
pinned only by its own round trip, by hand-assembled blocks in tests/test_host_cpu.py (prefix
compression, restart arrays, snappy-compressed blocks, masked CRC-32C known answers) and by the
LevelDB / snappy / CRC-32C specifications -- NOT by a file produced by TensorFlow.

Reader: footer -> index block -> data blocks (uncompressed or snappy) -> entries.  Writer:
uncompressed blocks (type 0), which every TF reader accepts; restart interval 16 and 256 KiB
blocks like TF's defaults.
"""
from __future__ import annotations

import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
MASK_DELTA = 0xa282ead8
BLOCK_SIZE = 262144
RESTART_INTERVAL = 16

# tensorflow/core/framework/types.proto
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64,
          10: np.bool_, 17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
DTYPE_IDS = {np.dtype(v): k for k, v in DTYPES.items()}


# ----------------------------------------------------------------------------- CRC-32C -----------
def _make_crc_table():
    t = np.zeros(256, np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        t[i] = c
    return t


_CRC_TABLE = _make_crc_table()
_native_crc = None


def _load_native_crc():
    """The C-ABI library carries a host-side CRC-32C (hardware instruction); use it when present."""
    global _native_crc
    if _native_crc is None:
        try:
            import ctypes as C
            from . import _lib as L
            lib = C.CDLL(L.LIB_PATH)
            lib.comic_crc32c.restype = C.c_uint32
            lib.comic_crc32c.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32]
            _native_crc = lib.comic_crc32c
        except Exception:           # CPU-only environments without the built library
            _native_crc = False
    return _native_crc


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli), the checksum of TF's table blocks and bundle entries."""
    buf = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data.view(np.uint8).reshape(-1)
    fn = _load_native_crc()
    if fn and buf.size:
        buf = np.ascontiguousarray(buf)
        return int(fn(buf.ctypes.data, buf.size, crc))
    c = np.uint32(crc ^ 0xFFFFFFFF)
    tab = _CRC_TABLE
    c = int(c)
    for b in buf.tobytes():
        c = int(tab[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(c):
    return (((c >> 15) | (c << 17)) + MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(m):
    r = (m - MASK_DELTA) & 0xFFFFFFFF
    return ((r >> 17) | (r << 15)) & 0xFFFFFFFF


# ----------------------------------------------------------------------------- varints / protos --
def _put_varint(out, v):
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)


def _get_varint(buf, pos):
    shift = v = 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if b < 0x80:
            return v, pos
        shift += 7


def _parse_proto(buf):
    """Minimal protobuf wire parser -> list of (field, wire_type, value)."""
    out, pos, n = [], 0, len(buf)
    while pos < n:
        key, pos = _get_varint(buf, pos)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        out.append((f, wt, v))
    return out


def _encode_header():
    """BundleHeaderProto{num_shards=1, endianness=LITTLE(0, default, omitted), version{producer=1}}."""
    out = bytearray()
    out += b'\x08\x01'                 # field 1 varint: num_shards = 1
    out += b'\x1a\x02\x08\x01'         # field 3 message VersionDef{producer = 1}
    return bytes(out)


def _encode_entry(dtype_id, shape, offset, size, crc_masked):
    """BundleEntryProto{dtype=1, shape=2, shard_id=3 (0, omitted), offset=4, size=5, crc32c=6 fixed32}."""
    out = bytearray()
    out.append(0x08)
    _put_varint(out, dtype_id)
    sh = bytearray()
    for d in shape:                    # TensorShapeProto{repeated Dim dim = 2 {int64 size = 1}}
        dim = bytearray(b'\x08')
        _put_varint(dim, int(d))
        sh.append(0x12)
        _put_varint(sh, len(dim))
        sh += dim
    out.append(0x12)
    _put_varint(out, len(sh))
    out += sh
    if offset:
        out.append(0x20)
        _put_varint(out, offset)
    out.append(0x28)
    _put_varint(out, size)
    out.append(0x35)
    out += struct.pack('<I', crc_masked)
    return bytes(out)


def _decode_entry(buf):
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=None, slices=0)
    for f, wt, v in _parse_proto(buf):
        if f == 1:
            e['dtype'] = v
        elif f == 2:
            for f2, _, v2 in _parse_proto(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _parse_proto(v2):
                        if f3 == 1:
                            size = v3 - (1 << 64) if v3 >> 63 else v3
                    e['shape'].append(size)
        elif f == 3:
            e['shard_id'] = v
        elif f == 4:
            e['offset'] = v
        elif f == 5:
            e['size'] = v
        elif f == 6:
            e['crc32c'] = v
        elif f == 7:
            e['slices'] += 1
    return e


# ----------------------------------------------------------------------------- snappy -------------
def snappy_uncompress(src):
    """Raw snappy block format (TF compresses table blocks with it by default)."""
    n, pos = _get_varint(src, 0)
    out = bytearray()
    ln = len(src)
    while pos < ln:
        tag = src[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            l = tag >> 2
            if l >= 60:
                nb = l - 59
                l = int.from_bytes(src[pos:pos + nb], 'little')
                pos += nb
            l += 1
            out += src[pos:pos + l]
            pos += l
            continue
        if kind == 1:
            l = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | src[pos]
            pos += 1
        elif kind == 2:
            l = (tag >> 2) + 1
            off = src[pos] | (src[pos + 1] << 8)
            pos += 2
        else:
            l = (tag >> 2) + 1
            off = int.from_bytes(src[pos:pos + 4], 'little')
            pos += 4
        if off == 0 or off > len(out):
            raise ValueError('corrupt snappy stream')
        for _ in range(l):                               # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError('snappy length mismatch (%d != %d)' % (len(out), n))
    return bytes(out)


# ----------------------------------------------------------------------------- table blocks -------
def _parse_block(block):
    """-> [(key, value)] of one table block (prefix-compressed entries + restart array)."""
    num_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * num_restarts
    out, pos, key = [], 0, b''
    while pos < end:
        shared, pos = _get_varint(block, pos)
        unshared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + unshared])
        pos += unshared
        out.append((key, bytes(block[pos:pos + vlen])))
        pos += vlen
    return out


def _read_block(data, offset, size, verify=True):
    raw = data[offset:offset + size]
    ctype = data[offset + size]
    stored = struct.unpack_from('<I', data, offset + size + 1)[0]
    if verify:
        actual = crc32c(bytes(data[offset:offset + size + 1]))
        if unmask_crc(stored) != actual:
            raise ValueError('table block checksum mismatch at offset %d' % offset)
    if ctype == 0:
        return bytes(raw)
    if ctype == 1:
        return snappy_uncompress(bytes(raw))
    raise ValueError('unknown block compression type %d' % ctype)


def _decode_handle(buf, pos=0):
    off, pos = _get_varint(buf, pos)
    size, pos = _get_varint(buf, pos)
    return off, size, pos


class _BlockBuilder:
    def __init__(self):
        self.buf, self.restarts, self.count, self.last = bytearray(), [0], 0, b''

    def add(self, key, value):
        shared = 0
        if self.count % RESTART_INTERVAL == 0 and self.count:
            self.restarts.append(len(self.buf))
        elif self.count:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        _put_varint(self.buf, shared)
        _put_varint(self.buf, len(key) - shared)
        _put_varint(self.buf, len(value))
        self.buf += key[shared:]
        self.buf += value
        self.last = key
        self.count += 1

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self):
        out = bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))
        return out


def _emit_block(f, contents):
    """Write block + trailer (type 0 = uncompressed, masked CRC-32C of contents + type) -> handle."""
    off = f.tell()
    f.write(contents)
    f.write(b'\x00')
    f.write(struct.pack('<I', mask_crc(crc32c(contents + b'\x00'))))
    h = bytearray()
    _put_varint(h, off)
    _put_varint(h, len(contents))
    return bytes(h)


def _write_table(path, items):
    """items: sorted [(key bytes, value bytes)]."""
    with open(path, 'wb') as f:
        index = _BlockBuilder()
        blk = _BlockBuilder()
        for key, value in items:
            blk.add(key, value)
            if blk.size() >= BLOCK_SIZE:
                index.add(blk.last, _emit_block(f, blk.finish()))     # separator = last key of the block
                blk = _BlockBuilder()
        if blk.count:
            index.add(blk.last, _emit_block(f, blk.finish()))
        meta_handle = _emit_block(f, _BlockBuilder().finish())
        index_handle = _emit_block(f, index.finish())
        footer = meta_handle + index_handle
        footer += b'\x00' * (40 - len(footer))
        footer += struct.pack('<Q', TABLE_MAGIC)
        f.write(footer)


def _read_table(path, verify=True):
    data = open(path, 'rb').read()
    if len(data) < 48 or struct.unpack_from('<Q', data, len(data) - 8)[0] != TABLE_MAGIC:
        raise ValueError('%s is not a TensorFlow table file (bad magic)' % path)
    footer = data[-48:]
    _, _, pos = _decode_handle(footer, 0)
    ioff, isize, _ = _decode_handle(footer, pos)
    out = []
    for _, handle in _parse_block(_read_block(data, ioff, isize, verify)):
        boff, bsize, _ = _decode_handle(handle)
        out += _parse_block(_read_block(data, boff, bsize, verify))
    return out


# ----------------------------------------------------------------------------- public API ---------
def data_path(prefix, shard=0, num_shards=1):
    return '%s.data-%05d-of-%05d' % (prefix, shard, num_shards)


def list_variables(prefix):
    """-> {name: (numpy dtype, shape tuple)}  (tf.train.list_variables)."""
    out = {}
    for key, value in _read_table(prefix + '.index'):
        if key == b'':
            continue
        e = _decode_entry(value)
        out[key.decode()] = (np.dtype(DTYPES[e['dtype']]), tuple(e['shape']))
    return out


def read_bundle(prefix, names=None, verify=True):
    """-> {variable name: numpy array}.  `names`: optional subset."""
    entries, num_shards = {}, 1
    for key, value in _read_table(prefix + '.index', verify):
        if key == b'':
            for f, _, v in _parse_proto(value):
                if f == 1:
                    num_shards = v
                if f == 2 and v != 0:
                    raise ValueError('big-endian bundles are not supported')
            continue
        entries[key.decode()] = _decode_entry(value)
    files = {}
    out = {}
    for name, e in entries.items():
        if names is not None and name not in names:
            continue
        if e['slices']:
            raise ValueError('partitioned variable %s is not supported' % name)
        if e['dtype'] not in DTYPES:
            raise ValueError('variable %s has unsupported dtype id %d' % (name, e['dtype']))
        sid = e['shard_id']
        if sid not in files:
            files[sid] = np.memmap(data_path(prefix, sid, num_shards), dtype=np.uint8, mode='r')
        raw = np.asarray(files[sid][e['offset']:e['offset'] + e['size']])
        if verify and e['crc32c'] is not None and unmask_crc(e['crc32c']) != crc32c(raw):
            raise ValueError('checksum mismatch for variable %s' % name)
        dt = np.dtype(DTYPES[e['dtype']])
        out[name] = raw.view(dt).reshape(e['shape']).copy()
    return out


def write_bundle(prefix, tensors):
    """tensors: {variable name: array}.  Writes <prefix>.index and <prefix>.data-00000-of-00001."""
    os.makedirs(os.path.dirname(prefix) or '.', exist_ok=True)
    items = [(b'', _encode_header())]
    off = 0
    with open(data_path(prefix), 'wb') as f:
        for name in sorted(tensors, key=lambda s: s.encode()):
            a = np.asarray(tensors[name])
            if not a.flags.c_contiguous:
                a = np.ascontiguousarray(a)
            if a.dtype not in DTYPE_IDS:
                raise ValueError('variable %s: dtype %s has no TF DataType here' % (name, a.dtype))
            raw = a.reshape(-1).view(np.uint8) if a.size else np.zeros(0, np.uint8)
            f.write(raw.tobytes())
            items.append((name.encode(), _encode_entry(DTYPE_IDS[a.dtype], a.shape, off, raw.size, mask_crc(crc32c(raw)))))
            off += raw.size
    _write_table(prefix + '.index', items)
    return prefix


def update_checkpoint_state(directory, prefix_basename, keep=None):
    """The `checkpoint` text proto tf.train.Saver maintains (latest_checkpoint reads it)."""
    path = os.path.join(directory, 'checkpoint')
    allp = []
    if os.path.isfile(path):
        for line in open(path):
            if line.startswith('all_model_checkpoint_paths:'):
                allp.append(line.split(':', 1)[1].strip().strip('"'))
    if prefix_basename in allp:
        allp.remove(prefix_basename)
    allp.append(prefix_basename)
    if keep:
        allp = allp[-keep:]
    with open(path, 'w') as f:
        f.write('model_checkpoint_path: "%s"\n' % prefix_basename)
        for p in allp:
            f.write('all_model_checkpoint_paths: "%s"\n' % p)
    return allp


def prune_checkpoint_state(directory, removed):
    """Drop deleted prefixes from `all_model_checkpoint_paths` (what tf.train.Saver does when max_to_keep evicts)."""
    path = os.path.join(directory, 'checkpoint')
    if not removed or not os.path.isfile(path):
        return
    lines = [l for l in open(path)
             if not (l.startswith('all_model_checkpoint_paths:') and l.split(':', 1)[1].strip().strip('"') in removed)]
    with open(path, 'w') as f:
        f.writelines(lines)


def latest_checkpoint(directory):
    """tf.train.latest_checkpoint: the prefix named by the `checkpoint` state file."""
    path = os.path.join(directory, 'checkpoint')
    if not os.path.isfile(path):
        return None
    for line in open(path):
        if line.startswith('model_checkpoint_path:'):
            p = line.split(':', 1)[1].strip().strip('"')
            p = p if os.path.isabs(p) else os.path.join(directory, p)
            return p if os.path.isfile(p + '.index') else None
    return None
