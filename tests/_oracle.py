"""ctypes bindings for the TEST-ONLY checkers: oracle/liboracle.so (our C restatement)
and oracle/_ref/libdsrc_ref.so (the unmodified reference, when it has been built).

Nothing under dsrc_amd/ may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


from dsrc_amd.config import Config      # noqa: E402,F401  (the settings record is the product's; the tests' modules import it from here)


class _OrcConfig(C.Structure):
    _fields_ = [("dna_order", C.c_uint32), ("quality_order", C.c_uint32), ("tag_preserve_flags", C.c_uint64),
                ("lossy", C.c_int32), ("calc_crc32", C.c_int32), ("quality_offset", C.c_uint32),
                ("plus_repetition", C.c_int32), ("color_space", C.c_int32)]


def _orc_cfg(cfg: Config) -> _OrcConfig:
    return _OrcConfig(cfg.dna_order, cfg.quality_order, cfg.tag_flags, int(cfg.lossy), int(cfg.crc),
                      cfg.quality_offset, int(cfg.plus_repetition), int(cfg.color_space))


def build_oracle() -> str:
    path = os.path.join(ORACLE_DIR, "liboracle.so")
    src = os.path.join(ORACLE_DIR, "dsrc_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "oracle"], stdout=subprocess.DEVNULL)
    return path


class Oracle:
    def __init__(self):
        self.lib = C.CDLL(build_oracle())
        L = self.lib
        L.orc_compress_block.restype = C.c_int
        L.orc_cut_chunks.restype = C.c_int64
        L.orc_archive_footer.restype = C.c_uint64
        L.orc_archive_header.restype = C.c_uint64
        for f in ("orc_bitwriter_script", "orc_huffman", "orc_rc_script", "orc_rc_adaptive4"):
            getattr(L, f).restype = C.c_uint64
        L.orc_crc32.restype = C.c_uint32

    def compress_block(self, cfg: Config, data: bytes):
        c = _orc_cfg(cfg)
        cap = len(data) + (1 << 16)
        out = (C.c_uint8 * cap)()
        osz = C.c_uint64(0)
        raw = (C.c_uint64 * 4)(); comp = (C.c_uint64 * 4)()
        rc = self.lib.orc_compress_block(C.byref(c), data, C.c_uint64(len(data)), out, C.c_uint64(cap),
                                         C.byref(osz), raw, comp)
        if rc != 0:
            raise RuntimeError(f"orc_compress_block rc={rc}")
        return bytes(out[: osz.value]), list(raw), list(comp)

    def compress_blocks_state(self, cfg: Config, chunks, fields_cap: int = 0):
        """Blocks of consecutive chunks from ONE BlockCompressor (`dsrc c -t1`): the capacity of TagStats::fields is
        carried from chunk to chunk.  Returns [(block, raw, comp)] and leaves the final capacity in self.last_fields_cap."""
        c = _orc_cfg(cfg)
        cap_state = C.c_uint32(fields_cap)
        res = []
        for data in chunks:
            cap = len(data) + (1 << 16)
            out = (C.c_uint8 * cap)(); osz = C.c_uint64(0)
            raw = (C.c_uint64 * 4)(); comp = (C.c_uint64 * 4)()
            rc = self.lib.orc_compress_block_state(C.byref(c), C.byref(cap_state), data, C.c_uint64(len(data)), out,
                                                   C.c_uint64(cap), C.byref(osz), raw, comp)
            if rc != 0:
                raise RuntimeError(f"orc_compress_block_state rc={rc}")
            res.append((bytes(out[: osz.value]), list(raw), list(comp)))
        self.last_fields_cap = cap_state.value
        return res

    def compress_records_block(self, cfg: Config, data: bytes, chunk_size: int, fields_cap: int = 0):
        """BlockCompressorExt::Flush for a chunk given as text; returns (block, new fields_cap)."""
        c = _orc_cfg(cfg)
        cap = len(data) + (1 << 16)
        out = (C.c_uint8 * cap)()
        osz = C.c_uint64(0); fc = C.c_uint32(fields_cap)
        raw = (C.c_uint64 * 4)(); comp = (C.c_uint64 * 4)()
        rc = self.lib.orc_compress_records_block(C.byref(c), C.byref(fc), C.c_uint32(chunk_size & 0xFFFFFFFF), data,
                                                 C.c_uint64(len(data)), out, C.c_uint64(cap), C.byref(osz), raw, comp)
        if rc != 0:
            raise RuntimeError(f"orc_compress_records_block rc={rc}")
        return bytes(out[: osz.value]), fc.value

    def compress_records_file(self, src: str, dst: str, d: int, q: int, lossy=False, qoff=33, buf_mb=8, plus_rep=False):
        return self.lib.orc_compress_records_file(src.encode(), dst.encode(), d, q, int(lossy), qoff, buf_mb, int(plus_rep))

    def block_stats(self, cfg: Config, data: bytes):
        c = _orc_cfg(cfg)
        d = (C.c_uint32 * 21)(); q = (C.c_uint32 * 262)()
        recs = C.c_uint64(); cs = C.c_uint64(); raw = (C.c_uint64 * 4)()
        rc = self.lib.orc_block_stats(C.byref(c), data, C.c_uint64(len(data)), d, q, C.byref(recs), C.byref(cs), raw)
        if rc != 0:
            raise RuntimeError(f"orc_block_stats rc={rc}")
        return list(d), list(q), recs.value, cs.value, list(raw)

    def analyze(self, data: bytes, estimate: bool = True, qoff: int = 0):
        off = C.c_uint32(qoff); pr = C.c_int32(); cs = C.c_int32()
        rc = self.lib.orc_analyze(data, C.c_uint64(len(data)), int(estimate), C.byref(off), C.byref(pr), C.byref(cs))
        return rc, off.value, bool(pr.value), bool(cs.value)

    def cut_chunks(self, data: bytes, buf_size: int):
        cap = len(data) // max(buf_size - 8192, 1) + 8
        st = (C.c_uint64 * cap)(); sz = (C.c_uint64 * cap)()
        n = self.lib.orc_cut_chunks(data, C.c_uint64(len(data)), C.c_uint64(buf_size), st, sz, C.c_uint64(cap))
        return [(st[i], sz[i]) for i in range(n)]

    def compress_file(self, src: str, dst: str, d: int, q: int, lossy=False, crc=False, qoff=0, buf_mb=8):
        return self.lib.orc_compress_file(src.encode(), dst.encode(), d, q, int(lossy), int(crc), qoff, buf_mb)

    def archive_footer(self, sizes, cfg: Config) -> bytes:
        c = _orc_cfg(cfg)
        arr = (C.c_uint32 * len(sizes))(*sizes)
        out = (C.c_uint8 * (len(sizes) * 4 + 32))()
        n = self.lib.orc_archive_footer(out, arr, C.c_uint64(len(sizes)), C.byref(c))
        return bytes(out[:n])

    def archive_header(self, footer_offset, footer_size, nblocks) -> bytes:
        out = (C.c_uint8 * 40)()
        self.lib.orc_archive_header(out, C.c_uint64(footer_offset), C.c_uint32(footer_size), C.c_uint64(nblocks))
        return bytes(out)

    # primitives ------------------------------------------------------
    def bitwriter_script(self, ops):
        flat = (C.c_uint32 * (3 * len(ops)))(*[x for op in ops for x in op])
        out = (C.c_uint8 * 4096)()
        n = self.lib.orc_bitwriter_script(flat, len(ops), out, C.c_uint64(4096))
        return bytes(out[:n])

    def huffman(self, freqs):
        n = len(freqs)
        f = (C.c_uint32 * n)(*freqs); codes = (C.c_uint32 * max(n, 2))(); lens = (C.c_uint32 * max(n, 2))()
        tree = (C.c_uint8 * 8192)()
        sz = self.lib.orc_huffman(f, n, codes, lens, tree, C.c_uint64(8192))
        return list(codes[:n]), list(lens[:n]), bytes(tree[:sz])

    def rc_script(self, fct):
        flat = (C.c_uint32 * (3 * len(fct)))(*[x for t in fct for x in t])
        cap = len(fct) * 4 + 64
        out = (C.c_uint8 * cap)()
        n = self.lib.orc_rc_script(flat, len(fct), out, C.c_uint64(cap))
        return bytes(out[:n])

    def rc_adaptive4(self, syms: bytes):
        cap = len(syms) + 64
        out = (C.c_uint8 * cap)()
        n = self.lib.orc_rc_adaptive4(syms, len(syms), out, C.c_uint64(cap))
        return bytes(out[:n])

    def crc32(self, data: bytes) -> int:
        return self.lib.orc_crc32(data, len(data))

    def decompress_block(self, cfg: Config, block: bytes, cap: int, with_crc: bool = False):
        """BlockCompressor::Read restated (oracle/dsrc_oracle_dec.c) -> chunk text incl. the final newline."""
        c = _orc_cfg(cfg)
        out = (C.c_uint8 * (cap + 64))()
        osz = C.c_uint64(0); st = (C.c_uint32 * 3)(); ac = (C.c_uint32 * 3)()
        rc = self.lib.orc_decompress_block(C.byref(c), block, C.c_uint64(len(block)), out, C.c_uint64(cap), C.byref(osz), st, ac)
        if rc != 0:
            raise RuntimeError(f"orc_decompress_block rc={rc}")
        text = bytes(out[: osz.value])
        return (text, list(st), list(ac)) if with_crc else text

    def verify_block(self, cfg: Config, block: bytes, cap: int) -> int:
        c = _orc_cfg(cfg)
        return self.lib.orc_verify_block(C.byref(c), block, C.c_uint64(len(block)), C.c_uint64(cap))


REF_SO = os.path.join(ORACLE_DIR, "_ref", "libdsrc_ref.so")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "dsrc_ref")


def have_ref() -> bool:
    return os.path.exists(REF_SO)


class Ref:
    """The unmodified reference (oracle/_ref, built by oracle/Makefile from /root/reference)."""

    def __init__(self):
        self.lib = C.CDLL(REF_SO)
        L = self.lib
        for f in ("ref_bitwriter_script", "ref_huffman", "ref_rc_script", "ref_rc_adaptive4"):
            getattr(L, f).restype = C.c_uint64
        L.ref_crc32.restype = C.c_uint32

    def compress_block(self, cfg: Config, data: bytes):
        cap = len(data) + (1 << 17)
        out = (C.c_uint8 * cap)()
        osz = C.c_uint64(0)
        raw = (C.c_uint64 * 4)(); comp = (C.c_uint64 * 4)()
        rc = self.lib.ref_compress_block(cfg.dna_order, cfg.quality_order, int(cfg.lossy), int(cfg.crc),
                                         C.c_uint64(cfg.tag_flags), cfg.quality_offset, int(cfg.plus_repetition),
                                         int(cfg.color_space), data, C.c_uint64(len(data)), out, C.c_uint64(cap),
                                         C.byref(osz), raw, comp)
        if rc != 0:
            raise RuntimeError(f"ref_compress_block rc={rc}")
        return bytes(out[: osz.value]), list(raw), list(comp)

    def decompress_block(self, cfg: Config, block: bytes, cap: int) -> bytes:
        out = (C.c_uint8 * (cap + 64))()
        osz = C.c_uint64(0)
        rc = self.lib.ref_decompress_block(cfg.dna_order, cfg.quality_order, int(cfg.lossy), int(cfg.crc),
                                           C.c_uint64(cfg.tag_flags), cfg.quality_offset, int(cfg.plus_repetition),
                                           int(cfg.color_space), block, C.c_uint64(len(block)), out,
                                           C.c_uint64(cap), C.byref(osz))
        if rc != 0:
            raise RuntimeError(f"ref_decompress_block rc={rc}")
        return bytes(out[: osz.value])

    def block_stats(self, cfg: Config, data: bytes):
        d = (C.c_uint32 * 21)(); q = (C.c_uint32 * 262)()
        recs = C.c_uint64(); cs = C.c_uint64(); raw = (C.c_uint64 * 4)()
        self.lib.ref_block_stats(int(cfg.lossy), cfg.quality_offset, data, C.c_uint64(len(data)), d, q,
                                 C.byref(recs), C.byref(cs), raw)
        return list(d), list(q), recs.value, cs.value, list(raw)

    def analyze(self, data: bytes, estimate: bool = True, qoff: int = 0):
        off = C.c_uint32(qoff); pr = C.c_int(); cs = C.c_int()
        rc = self.lib.ref_analyze(data, C.c_uint64(len(data)), int(estimate), C.byref(off), C.byref(pr), C.byref(cs))
        return rc, off.value, bool(pr.value), bool(cs.value)

    def chunk_sizes(self, path: str, buf_mb: int):
        cap = 1 << 16
        sz = (C.c_uint64 * cap)(); n = C.c_uint32()
        self.lib.ref_chunk_sizes(path.encode(), buf_mb, sz, cap, C.byref(n))
        return [sz[i] for i in range(n.value)]

    def compress_file(self, src, dst, d, q, lossy=False, crc=False, qoff=0, buf_mb=8, threads=1):
        return self.lib.ref_compress_file(src.encode(), dst.encode(), d, q, int(lossy), int(crc), qoff, buf_mb,
                                          threads, C.c_uint64(0))

    def decompress_file(self, src, dst, threads=1):
        return self.lib.ref_decompress_file(src.encode(), dst.encode(), threads)

    def bitwriter_script(self, ops):
        flat = (C.c_uint32 * (3 * len(ops)))(*[x for op in ops for x in op])
        out = (C.c_uint8 * 4096)()
        n = self.lib.ref_bitwriter_script(flat, len(ops), out, C.c_uint64(4096))
        return bytes(out[:n])

    def huffman(self, freqs):
        n = len(freqs)
        f = (C.c_uint32 * n)(*freqs); codes = (C.c_uint32 * max(n, 2))(); lens = (C.c_uint32 * max(n, 2))()
        tree = (C.c_uint8 * 8192)()
        sz = self.lib.ref_huffman(f, n, codes, lens, tree, C.c_uint64(8192))
        return list(codes[:n]), list(lens[:n]), bytes(tree[:sz])

    def rc_script(self, fct):
        flat = (C.c_uint32 * (3 * len(fct)))(*[x for t in fct for x in t])
        cap = len(fct) * 4 + 64
        out = (C.c_uint8 * cap)()
        n = self.lib.ref_rc_script(flat, len(fct), out, C.c_uint64(cap))
        return bytes(out[:n])

    def rc_adaptive4(self, syms: bytes):
        cap = len(syms) + 64
        out = (C.c_uint8 * cap)()
        n = self.lib.ref_rc_adaptive4(syms, len(syms), out, C.c_uint64(cap))
        return bytes(out[:n])

    def crc32(self, data: bytes) -> int:
        return self.lib.ref_crc32(data, len(data))
