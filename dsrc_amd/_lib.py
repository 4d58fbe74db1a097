"""ctypes binding of libdsrc_gpu.so (include/dsrc_gpu.h).

The product path has exactly one implementation -- the HIP library.  If it has not been built, or
there is no GPU, loading / creating a handle fails loudly; nothing here falls back to a CPU codec.

``DSRC_GPU_LIB`` may point at another build of the same C ABI (the test-suite uses it to load the
emulator build under tests/emu for kernel-logic tests on GPU-less machines).
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "csrc", "libdsrc_gpu.so")

EXPORTS = [
    "dsrcgpu_create", "dsrcgpu_destroy", "dsrcgpu_last_error", "dsrcgpu_compress_block", "dsrcgpu_compress_batch",
    "dsrcgpu_compress_batch_device", "dsrcgpu_submit", "dsrcgpu_flush", "dsrcgpu_collect", "dsrcgpu_release",
    "dsrcgpu_last_timing", "dsrcgpu_synth_illumina", "dsrcgpu_dev_alloc", "dsrcgpu_dev_free", "dsrcgpu_dev_upload",
    "dsrcgpu_dev_download", "dsrcgpu_chain_create", "dsrcgpu_chain_destroy", "dsrcgpu_set_chain", "dsrcgpu_host_alloc",
    "dsrcgpu_host_free", "dsrcgpu_selftest", "dsrcgpu_set_record_layout",
    "dsrcgpu_decompress_block", "dsrcgpu_decompress_batch", "dsrcgpu_decompress_batch_device",
    "dsrcgpu_title_fields", "dsrcgpu_fields_capacity_after", "dsrcgpu_set_fields_capacity", "dsrcgpu_get_fields_capacity",
    "dsrcgpu_chain_seed", "dsrcgpu_last_stage_timing", "dsrcgpu_try_collect", "dsrcgpu_prepare", "dsrcgpu_set_table_budget", "dsrcgpu_device_memory", "dsrcgpu_release_memory",
    "dsrcgpu_synth_fastq", "dsrcgpu_reserve_memory", "dsrcgpu_set_lanes", "dsrcgpu_submit_pinned",
]


class Settings(C.Structure):
    _fields_ = [("dna_order", C.c_uint32), ("quality_order", C.c_uint32), ("tag_preserve_flags", C.c_uint64),
                ("lossy", C.c_uint8), ("calculate_crc32", C.c_uint8), ("verify_after_compress", C.c_uint8),
                ("reserved", C.c_uint8 * 5)]


class Dataset(C.Structure):
    _fields_ = [("quality_offset", C.c_uint32), ("plus_repetition", C.c_uint8), ("color_space", C.c_uint8),
                ("reserved", C.c_uint8 * 2)]


class DsrcGpuError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"dsrc_gpu error {code}: {msg}")
        self.code = code


_lib = None


def lib_path() -> str:
    return os.environ.get("DSRC_GPU_LIB", DEFAULT_LIB)


def load():
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          f"(hipcc --offload-arch=gfx950). The DSRC GPU path has no CPU fallback.")
    L = C.CDLL(path)
    L.dsrcgpu_last_error.restype = C.c_char_p
    L.dsrcgpu_last_error.argtypes = [C.c_void_p]
    L.dsrcgpu_destroy.restype = None
    L.dsrcgpu_destroy.argtypes = [C.c_void_p]
    L.dsrcgpu_chain_destroy.restype = None
    L.dsrcgpu_chain_destroy.argtypes = [C.c_void_p]
    L.dsrcgpu_set_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.dsrcgpu_set_record_layout.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.dsrcgpu_title_fields.restype = C.c_uint32
    L.dsrcgpu_title_fields.argtypes = [C.c_char_p, C.c_uint32, C.c_uint64]
    L.dsrcgpu_fields_capacity_after.restype = C.c_uint32
    L.dsrcgpu_fields_capacity_after.argtypes = [C.c_uint32, C.c_uint32]
    L.dsrcgpu_chain_seed.argtypes = [C.c_void_p, C.c_uint32]
    _lib = L
    return L


def fields_capacity_fold(chunks, tag_flags: int = 0, cap: int = 0) -> int:
    """The block-to-block state (capacity of the reference's TagStats::fields) after `chunks` have been compressed in
    order, from the first title line of each chunk alone (include/dsrc_gpu.h: dsrcgpu_fields_capacity_after)."""
    L = load()
    for c in chunks:
        e = 0
        while e < len(c) and c[e] not in (10, 13):
            e += 1
        cap = L.dsrcgpu_fields_capacity_after(cap, L.dsrcgpu_title_fields(bytes(c[:e]), e, tag_flags))
    return cap


def host_alloc(nbytes: int) -> int:
    """Page-locked host memory (dsrcgpu_host_alloc); free with host_free."""
    p = C.c_void_p()
    rc = load().dsrcgpu_host_alloc(C.c_uint64(nbytes), C.byref(p))
    if rc:
        raise DsrcGpuError(rc, "dsrcgpu_host_alloc failed")
    return p.value


def host_free(ptr: int):
    load().dsrcgpu_host_free(C.c_void_p(ptr))


class Chain:
    """Hands DSRC's block-to-block state from batch seq to batch seq+1 across handles (include/dsrc_gpu.h)."""

    def __init__(self):
        self.L = load()
        self.c = C.c_void_p()
        rc = self.L.dsrcgpu_chain_create(C.byref(self.c))
        if rc != 0:
            raise DsrcGpuError(rc, "dsrcgpu_chain_create failed")

    def seed(self, fields_capacity: int):
        rc = self.L.dsrcgpu_chain_seed(self.c, fields_capacity)
        if rc != 0:
            raise DsrcGpuError(rc, "dsrcgpu_chain_seed: the chain has already started")

    def close(self):
        if getattr(self, "c", None):
            self.L.dsrcgpu_chain_destroy(self.c)
            self.c = None


class Handle:
    """One GPU block scheduler (replaces the reference's pool of BlockCompressor worker threads)."""

    def __init__(self, dna_order=0, quality_order=0, lossy=False, crc=False, quality_offset=33,
                 plus_repetition=False, color_space=False, tag_flags=0, device=0, arena_bytes=0, verify=False):
        self.L = load()
        self.h = C.c_void_p()
        self.device = device
        s = Settings(dna_order, quality_order, tag_flags, int(lossy), int(crc), int(verify))
        d = Dataset(quality_offset, int(plus_repetition), int(color_space))
        rc = self.L.dsrcgpu_create(C.byref(s), C.byref(d), device, C.c_uint64(arena_bytes), C.byref(self.h))
        if rc != 0:
            msg = self.L.dsrcgpu_last_error(self.h).decode() if self.h else "create failed"
            if self.h:
                self.L.dsrcgpu_destroy(self.h)
                self.h = None
            raise DsrcGpuError(rc, msg)

    def close(self):
        if getattr(self, "h", None):
            self.L.dsrcgpu_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise DsrcGpuError(rc, self.L.dsrcgpu_last_error(self.h).decode())
        return rc

    def set_chain(self, chain, seq: int):
        """The next batch call on this handle is batch number `seq` of `chain` (None detaches)."""
        self._chk(self.L.dsrcgpu_set_chain(self.h, chain.c if chain is not None else None, C.c_uint64(seq)))

    def set_lanes(self, lanes: int = 0, sub_batch_chunks: int = 0):
        """Scheduler lanes inside the handle (include/dsrc_gpu.h): 0 = the defaults, lanes = 1: batches stay on the handle's own lane."""
        self._chk(self.L.dsrcgpu_set_lanes(self.h, C.c_uint32(lanes), C.c_uint32(sub_batch_chunks)))

    def set_fields_capacity(self, cap: int):
        """Seed the block-to-block state of a handle that starts in the middle of an archive (see fields_capacity_fold)."""
        self._chk(self.L.dsrcgpu_set_fields_capacity(self.h, C.c_uint32(cap)))

    def get_fields_capacity(self) -> int:
        v = C.c_uint32()
        self._chk(self.L.dsrcgpu_get_fields_capacity(self.h, C.byref(v)))
        return v.value

    def set_record_layout(self, chunk_sizes):
        """The next batch call compresses chunks assembled from records (reference: BlockCompressorExt);
        chunk_sizes[i] is block i's chunkSize word."""
        arr = (C.c_uint32 * len(chunk_sizes))(*[v & 0xFFFFFFFF for v in chunk_sizes])
        self._chk(self.L.dsrcgpu_set_record_layout(self.h, len(chunk_sizes), arr))

    def compress_block(self, data: bytes):
        cap = len(data) + (1 << 16)
        out = (C.c_uint8 * cap)()
        osz = C.c_uint64()
        raw = (C.c_uint64 * 4)(); comp = (C.c_uint64 * 4)()
        self._chk(self.L.dsrcgpu_compress_block(self.h, data, C.c_uint64(len(data)), out, C.c_uint64(cap), C.byref(osz), raw, comp))
        return bytes(out[: osz.value]), list(raw), list(comp)

    def compress_batch(self, chunks, cap=None):
        n = len(chunks)
        ptrs = (C.c_char_p * n)(*chunks)
        sizes = (C.c_uint64 * n)(*[len(c) for c in chunks])
        if cap is None:
            cap = sum(len(c) for c in chunks) + n * (1 << 16)
        out = (C.c_uint8 * cap)()
        offs = (C.c_uint64 * n)(); osz = (C.c_uint64 * n)()
        raw = (C.c_uint64 * (4 * n))(); comp = (C.c_uint64 * (4 * n))()
        self._chk(self.L.dsrcgpu_compress_batch(self.h, n, ptrs, sizes, out, C.c_uint64(cap), offs, osz, raw, comp))
        mv = memoryview(out)
        return [(bytes(mv[offs[i]: offs[i] + osz[i]]), list(raw[4 * i: 4 * i + 4]), list(comp[4 * i: 4 * i + 4])) for i in range(n)]

    def decompress_batch(self, blocks, cap=None, text_caps=None, verify=False):
        """BlockCompressor::Read for a batch of blocks -> list of chunk texts (each ends with a newline);
        with verify=True also the per-block VerifyChecksum verdicts."""
        n = len(blocks)
        ptrs = (C.c_char_p * n)(*blocks)
        sizes = (C.c_uint64 * n)(*[len(b) for b in blocks])
        if cap is None:
            cap = sum(text_caps) if text_caps else sum(int.from_bytes(b[12:16], "big") + 1 for b in blocks)
        out = (C.c_uint8 * max(cap, 1))()
        offs = (C.c_uint64 * n)(); osz = (C.c_uint64 * n)(); ok = (C.c_uint32 * n)()
        caps = (C.c_uint64 * n)(*text_caps) if text_caps else None
        self._chk(self.L.dsrcgpu_decompress_batch(self.h, n, ptrs, sizes, caps, out, C.c_uint64(cap), offs, osz, ok if verify else None))
        mv = memoryview(out)
        texts = [bytes(mv[offs[i]: offs[i] + osz[i]]) for i in range(n)]
        return (texts, list(ok)) if verify else texts

    def decompress_batch_device(self, d_in: int, offs, sizes, d_out: int, out_cap: int, verify=False):
        n = len(offs)
        a_offs = (C.c_uint64 * n)(*offs); a_sizes = (C.c_uint64 * n)(*sizes)
        o_offs = (C.c_uint64 * n)(); o_sizes = (C.c_uint64 * n)(); ok = (C.c_uint32 * n)()
        self._chk(self.L.dsrcgpu_decompress_batch_device(self.h, n, C.c_void_p(d_in), a_offs, a_sizes, None, C.c_void_p(d_out),
                                                         C.c_uint64(out_cap), o_offs, o_sizes, ok if verify else None))
        return (list(o_offs), list(o_sizes), list(ok)) if verify else (list(o_offs), list(o_sizes))

    def compress_batch_device(self, d_in: int, offs, sizes, d_out: int, out_cap: int):
        n = len(offs)
        a_offs = (C.c_uint64 * n)(*offs); a_sizes = (C.c_uint64 * n)(*sizes)
        o_offs = (C.c_uint64 * n)(); o_sizes = (C.c_uint64 * n)()
        raw = (C.c_uint64 * (4 * n))(); comp = (C.c_uint64 * (4 * n))()
        self._chk(self.L.dsrcgpu_compress_batch_device(self.h, n, C.c_void_p(d_in), a_offs, a_sizes, C.c_void_p(d_out),
                                                       C.c_uint64(out_cap), o_offs, o_sizes, raw, comp))
        return list(o_offs), list(o_sizes), list(raw), list(comp)

    def submit_pinned(self, part_id: int, ptr: int, size: int) -> bool:
        """dsrcgpu_submit_pinned: the chunk at `ptr` (page-locked memory from host_alloc) is read in place; False = the ring is full."""
        rc = self.L.dsrcgpu_submit_pinned(self.h, C.c_int64(part_id), C.c_void_p(ptr), C.c_uint64(size))
        if rc == -8:
            return False
        self._chk(rc)
        return True

    def submit(self, part_id: int, data: bytes) -> bool:
        """False = all batches of the ring are in flight (DSRCGPU_E_BUSY): collect blocks, then submit again."""
        rc = self.L.dsrcgpu_submit(self.h, C.c_int64(part_id), data, C.c_uint64(len(data)))
        if rc == -8:
            return False
        self._chk(rc)
        return True

    def flush(self):
        self._chk(self.L.dsrcgpu_flush(self.h))

    def collect(self, wait: bool = True):
        pid = C.c_int64(); blk = C.POINTER(C.c_uint8)(); sz = C.c_uint64()
        raw = (C.c_uint64 * 4)(); comp = (C.c_uint64 * 4)()
        fn = self.L.dsrcgpu_collect if wait else self.L.dsrcgpu_try_collect
        rc = self._chk(fn(self.h, C.byref(pid), C.byref(blk), C.byref(sz), raw, comp))
        if rc == 0:
            return None
        data = bytes(C.cast(blk, C.POINTER(C.c_uint8 * sz.value)).contents) if sz.value else b""
        self.L.dsrcgpu_release(self.h, blk)
        return pid.value, data, list(raw), list(comp)

    def selftest(self) -> int:
        n = C.c_uint32(1)
        self._chk(self.L.dsrcgpu_selftest(self.h, C.byref(n)))
        return n.value

    def release_memory(self):
        """Hands the batch arena and the table region back to the device (they are allocated again on demand)."""
        self._chk(self.L.dsrcgpu_release_memory(self.h))

    def reserve_memory(self, arena_bytes, table_bytes=0):
        """Grows the batch arena / the decoder's table region to at least these sizes now (nothing shrinks)."""
        self._chk(self.L.dsrcgpu_reserve_memory(self.h, C.c_uint64(arena_bytes), C.c_uint64(table_bytes)))

    def last_timing(self):
        ms = C.c_float(); rc_ms = C.c_float(); n = C.c_uint32()
        self.L.dsrcgpu_last_timing(self.h, C.byref(ms), C.byref(rc_ms), C.byref(n))
        return ms.value, rc_ms.value, n.value

    def last_stage_timing(self):
        a = C.c_float(); b = C.c_float()
        self.L.dsrcgpu_last_stage_timing(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    # device helpers -------------------------------------------------
    def dev_alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self._chk(self.L.dsrcgpu_dev_alloc(self.h, C.c_uint64(nbytes), C.byref(p)))
        return p.value

    def dev_free(self, ptr: int):
        self._chk(self.L.dsrcgpu_dev_free(self.h, C.c_void_p(ptr)))

    def dev_upload(self, d_dst: int, data: bytes):
        self._chk(self.L.dsrcgpu_dev_upload(self.h, C.c_void_p(d_dst), data, C.c_uint64(len(data))))

    def dev_download(self, d_src: int, nbytes: int) -> bytes:
        buf = (C.c_uint8 * nbytes)()
        self._chk(self.L.dsrcgpu_dev_download(self.h, buf, C.c_void_p(d_src), C.c_uint64(nbytes)))
        return bytes(buf)

    def synth_illumina(self, first: int, count: int, d_out: int, cap: int, binned: bool = False) -> int:
        n = C.c_uint64()
        self._chk(self.L.dsrcgpu_synth_fastq(self.h, C.c_uint32(1 if binned else 0), C.c_uint64(first), C.c_uint64(count), C.c_void_p(d_out), C.c_uint64(cap), C.byref(n)))
        return n.value
