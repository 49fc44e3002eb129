"""Read-only access to an LMDB environment (``data.mdb``) without the ``lmdb`` package.

The reference's drivers read their feature stores through ``lmdb.open(path, readonly=True, ...).begin(buffers=True).get(key)``
(eval_nlq_retrieval_e2e2.py:187-192,238-255; eval_nlq_negative.py:147-152,203-221).  The package is not part of this image, so the one
operation the path needs - a point lookup in the main database of a single-writer, finished environment - is implemented here against the
PUBLISHED on-disk format of LMDB 0.9 (OpenLDAP ``lmdb.h`` / ``mdb.c``: ``MDB_meta``, ``MDB_page``, ``MDB_node``; data-format version 1):

  * pages of ``psize`` bytes (``mm_dbs[FREE_DBI].md_pad`` of the meta page; 4096 by default); page header 16 bytes:
    ``mp_pgno`` u64, ``mp_pad`` u16, ``mp_flags`` u16 (P_BRANCH 0x01, P_LEAF 0x02, P_OVERFLOW 0x04, P_META 0x08), then ``mp_lower`` /
    ``mp_upper`` u16 - or, on an overflow page, ``mp_pages`` u32: the number of pages of the run;
  * pages 0 and 1 are meta pages: at byte 16 ``mm_magic`` u32 = 0xBEEFC0DE, ``mm_version`` u32 = 1, ``mm_address`` u64, ``mm_mapsize`` u64,
    ``mm_dbs[2]`` (48 bytes each: ``md_pad`` u32, ``md_flags`` u16, ``md_depth`` u16, ``md_branch_pages`` / ``md_leaf_pages`` /
    ``md_overflow_pages`` / ``md_entries`` / ``md_root`` u64), ``mm_last_pg`` u64, ``mm_txnid`` u64; the meta page with the larger
    ``mm_txnid`` is the current one; ``mm_dbs[1]`` is the main database, ``md_root`` = 2^64 - 1 when it is empty;
  * a branch / leaf page holds ``(mp_lower - 16) / 2`` nodes; ``mp_ptrs[i]`` (u16 at 16 + 2 i) is the page offset of node i; nodes are
    sorted by key (memcmp, the shorter key first on a tie - the default comparator);
  * node: ``mn_lo`` u16, ``mn_hi`` u16, ``mn_flags`` u16, ``mn_ksize`` u16, the key, then: on a BRANCH page nothing (the child page number is
    ``mn_lo | mn_hi << 16 | mn_flags << 32``; node 0 has an empty key and stands for everything below node 1's key); on a LEAF page the
    data, ``mn_lo | mn_hi << 16`` bytes - or, with F_BIGDATA (0x01) set, the u64 page number of an OVERFLOW run whose bytes start 16 bytes
    into its first page and run on contiguously (that is where multi-megabyte ``np.savez_compressed`` feature blobs live).

Not supported (nothing on the path uses them): named sub-databases (F_SUBDATA), sorted duplicates (F_DUPDATA / P_LEAF2), custom
comparators, environments with an open writer, big-endian files.  No file written by liblmdb itself was available in the build container
(the package is absent), so this reader is checked against files laid out from the same published structures by the test suite's own
writer (tests/test_next_rows.py) - parity with liblmdb is therefore UNPINNED, and ``FeatureStore`` prefers the real package when it can
be imported.
"""
import io
import mmap
import os
import struct

P_BRANCH, P_LEAF, P_OVERFLOW, P_META = 0x01, 0x02, 0x04, 0x08
F_BIGDATA, F_SUBDATA, F_DUPDATA = 0x01, 0x02, 0x04
MAGIC, VERSION, PAGEHDR = 0xBEEFC0DE, 1, 16
P_INVALID = (1 << 64) - 1
_META = struct.Struct("<IIQQ")            # magic, version, address, mapsize
_DB = struct.Struct("<IHHQQQQQ")          # pad, flags, depth, branch, leaf, overflow, entries, root
_NODE = struct.Struct("<HHHH")            # lo, hi, flags, ksize


class MdbError(OSError):
    pass


class Environment:
    """``Environment(path)`` ~ ``lmdb.open(path, readonly=True, lock=False)``; ``begin()`` returns the environment itself, whose ``get``
    is the transaction's: ``env.begin(buffers=True).get(key)`` -> a ``memoryview`` of the value (into the mapping) or ``None``."""

    def __init__(self, path, subdir=True):
        self.path = os.path.join(path, "data.mdb") if (subdir and os.path.isdir(path)) else path
        self._f = io.open(self.path, "rb")        # (the module-level ``open`` below is lmdb's)
        size = os.fstat(self._f.fileno()).st_size
        if size < 2 * 512:
            raise MdbError(f"{self.path}: too small for an LMDB environment")
        self._m = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        self._v = memoryview(self._m)
        # the page size is recorded in meta page 0 (mm_dbs[0].md_pad); meta page 1 sits one page further
        first = self._meta_at(0)
        self.psize = first["psize"]
        if self.psize < 512 or self.psize & (self.psize - 1) or 2 * self.psize > size:
            raise MdbError(f"{self.path}: implausible page size {self.psize}")
        metas = [first, self._meta_at(self.psize)]
        self.meta = max(metas, key=lambda m: m["txnid"])
        self.root, self.depth, self.entries = self.meta["main"][7], self.meta["main"][2], self.meta["main"][6]
        if self.meta["main"][1] != 0:          # MDB_REVERSEKEY / MDB_DUPSORT / MDB_INTEGERKEY ... change the key order or the node format
            raise MdbError(f"{self.path}: main database flags {self.meta['main'][1]:#x} (custom key order / duplicates) are not supported")

    def _meta_at(self, off):
        flags = struct.unpack_from("<H", self._v, off + 10)[0]
        magic, version, _addr, mapsize = _META.unpack_from(self._v, off + PAGEHDR)
        if not flags & P_META or magic != MAGIC:
            raise MdbError(f"{self.path}: no LMDB meta page at byte {off} (magic {magic:#x})")
        if version != VERSION:
            raise MdbError(f"{self.path}: LMDB data format version {version} (this reader knows version {VERSION})")
        free = _DB.unpack_from(self._v, off + PAGEHDR + _META.size)
        main = _DB.unpack_from(self._v, off + PAGEHDR + _META.size + _DB.size)
        last_pg, txnid = struct.unpack_from("<QQ", self._v, off + PAGEHDR + _META.size + 2 * _DB.size)
        return {"psize": free[0], "main": main, "last_pg": last_pg, "txnid": txnid, "mapsize": mapsize}

    # ---- the lmdb surface the drivers use --------------------------------------------------------------------------------------
    def begin(self, *a, **kw):
        if kw.get("write"):
            raise MdbError("read-only environment")
        return self

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def close(self):
        self._v.release()
        self._m.close()
        self._f.close()

    def stat(self):
        return {"psize": self.psize, "depth": self.depth, "entries": self.entries, "last_pgno": self.meta["last_pg"], "last_txnid": self.meta["txnid"]}

    def _node(self, page_off, i):
        ptr = struct.unpack_from("<H", self._v, page_off + PAGEHDR + 2 * i)[0]
        lo, hi, flags, ksize = _NODE.unpack_from(self._v, page_off + ptr)
        k0 = page_off + ptr + _NODE.size
        return lo, hi, flags, self._v[k0:k0 + ksize], k0 + ksize

    @staticmethod
    def _less(a, b):
        """the default key order: memcmp over the common length, then the shorter key first"""
        return bytes(a) < bytes(b)

    def get(self, key, default=None):
        key = bytes(key)
        if self.root == P_INVALID:
            return default
        pgno = self.root
        for _ in range(64):          # (a B+tree over 2^64 pages is not 64 levels deep: a loop in a damaged file ends here)
            off = pgno * self.psize
            if off + self.psize > len(self._v):
                raise MdbError(f"{self.path}: page {pgno} lies outside the file")
            flags, lower = struct.unpack_from("<HH", self._v, off + 10)
            n = (lower - PAGEHDR) // 2
            if flags & P_BRANCH:
                # the last node whose key is <= the search key (node 0: empty key = minus infinity)
                lo_i, hi_i = 0, n - 1
                while lo_i < hi_i:
                    mid = (lo_i + hi_i + 1) // 2
                    if self._less(key, self._node(off, mid)[3]):
                        hi_i = mid - 1
                    else:
                        lo_i = mid
                lo, hi, fl, _, _ = self._node(off, lo_i)
                pgno = lo | (hi << 16) | (fl << 32)
                continue
            if not flags & P_LEAF or flags & 0x20:
                raise MdbError(f"{self.path}: page {pgno} has flags {flags:#x} (neither a branch nor a plain leaf page)")
            lo_i, hi_i = 0, n - 1
            while lo_i <= hi_i:
                mid = (lo_i + hi_i) // 2
                lo, hi, fl, k, d0 = self._node(off, mid)
                kb = bytes(k)
                if kb == key:
                    if fl & (F_SUBDATA | F_DUPDATA):
                        raise MdbError(f"{self.path}: key {key!r} holds a sub-database / duplicates (not supported)")
                    size = lo | (hi << 16)
                    if fl & F_BIGDATA:
                        opg = struct.unpack_from("<Q", self._v, d0)[0]
                        ooff = opg * self.psize
                        oflags, = struct.unpack_from("<H", self._v, ooff + 10)
                        npages, = struct.unpack_from("<I", self._v, ooff + 12)
                        if not oflags & P_OVERFLOW or PAGEHDR + size > npages * self.psize:
                            raise MdbError(f"{self.path}: damaged overflow run at page {opg}")
                        return self._v[ooff + PAGEHDR: ooff + PAGEHDR + size]
                    return self._v[d0:d0 + size]
                if kb < key:
                    lo_i = mid + 1
                else:
                    hi_i = mid - 1
            return default
        raise MdbError(f"{self.path}: B+tree deeper than 64 levels (damaged file)")


def open(path, readonly=True, create=False, **kw):  # noqa: A001 - lmdb's name
    """``lmdb.open`` for readers: the keyword arguments the reference passes (``max_readers``, ``readahead``, ...) are accepted and ignored."""
    if not readonly or create:
        raise MdbError("mdb_reader opens existing environments read-only")
    return Environment(path, subdir=kw.get("subdir", True))
