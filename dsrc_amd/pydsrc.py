"""`pydsrc`-compatible compression front end (reference: py/Interface.cpp:55-109, boost::python over
DsrcModule).  Same class and property names, so a script written for the reference's module runs unchanged for
compression:

    from dsrc_amd import pydsrc
    m = pydsrc.DsrcModule()
    m.DNACompressionLevel = 3; m.QualityCompressionLevel = 2
    m.Compress("in.fastq", "out.dsrc")

It drives the C++ host pipeline (`dsrc-amd`, dsrc_amd/csrc/host) on top of the C ABI; the archive is the one
`dsrc c -t1` writes.  FastqRecord / FastqFile / FieldMask / DsrcArchive give the record-level API for writing
archives.  Decompression (DsrcModule.Decompress, DsrcArchive.StartDecompress) is not part of the MI355X
path and raises RuntimeError, like the reference's module does for any DsrcException.

Note: the reference binds the *setter* of QualityCompressionLevel to SetDnaCompressionLevel (py/Interface.cpp:88,103),
so assigning it there silently changes the DNA level; here the property sets the quality level.
"""
from __future__ import annotations

import os
import subprocess

_CLI = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "dsrc-amd")


class DsrcModule:
    def __init__(self):
        self.LossyCompression = False
        self._dna = 0
        self._qua = 0
        self.TagFieldFilterMask = 0
        self._buf = 8
        self._threads = 4
        self.Crc32Checking = False
        self.Device = 0                     # GPU ordinal (not in the reference)

    # range checks of Configurable's setters (reference src/Configurable.cpp:56-179)
    @property
    def DNACompressionLevel(self):
        return self._dna

    @DNACompressionLevel.setter
    def DNACompressionLevel(self, v):
        if not 0 <= int(v) <= 3:
            raise RuntimeError("Invalid DNA compression mode specified [0-3]")
        self._dna = int(v)

    @property
    def QualityCompressionLevel(self):
        return self._qua

    @QualityCompressionLevel.setter
    def QualityCompressionLevel(self, v):
        if not 0 <= int(v) <= 2:
            raise RuntimeError("Invalid Quality compression mode specified [0-2]")
        self._qua = int(v)

    @property
    def FastqBufferSizeMB(self):
        return self._buf

    @FastqBufferSizeMB.setter
    def FastqBufferSizeMB(self, v):
        if not 1 <= int(v) <= 1024:
            raise RuntimeError("Invalid fastq buffer size specified [1-1024]")
        self._buf = int(v)

    @property
    def ThreadsNumber(self):
        return self._threads

    @ThreadsNumber.setter
    def ThreadsNumber(self, v):
        if not 1 <= int(v) <= 64:
            raise RuntimeError("Invalid thread number specified [1-64]")
        self._threads = int(v)

    def Compress(self, inputFilename: str, outputFilename: str) -> None:
        mask = int(self.TagFieldFilterMask)
        if mask & ~0x7FFFFFFE:
            raise RuntimeError("TagFieldFilterMask: field numbers 1..30 only")
        if not os.path.exists(_CLI):
            raise RuntimeError(f"{_CLI} not built: python -c 'import __graft_entry__ as g; g.build()'")
        cmd = [_CLI, "c", f"-d{self._dna}", f"-q{self._qua}", f"-b{self._buf}", f"-t{min(self._threads, 8)}", f"-g{int(self.Device)}"]
        if self.LossyCompression:
            cmd.append("-l")
        if self.Crc32Checking:
            cmd.append("-c")
        if mask:
            cmd.append("-f" + ",".join(str(k) for k in range(1, 31) if mask >> k & 1))
        r = subprocess.run(cmd + [inputFilename, outputFilename], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr.strip() or "dsrc-amd failed")

    def Decompress(self, inputFilename: str, outputFilename: str) -> None:
        raise RuntimeError("Decompression is not part of the MI355X hot path (SURVEY 8f-1); use the reference's DsrcModule.Decompress")


class FastqRecord:
    """py/Interface.cpp:59-64"""
    __slots__ = ("tag", "sequence", "plus", "quality")

    def __init__(self):
        self.tag = self.sequence = self.plus = self.quality = ""


class FieldMask:
    """include/dsrc/Configurable.h:22-43; AddField returns a new mask, as in the reference."""

    def __init__(self, mask: int = 0):
        self._mask = mask

    def AddField(self, i: int) -> "FieldMask":
        return FieldMask(self._mask | (1 << int(i)))

    def GetMask(self) -> int:
        return self._mask


class FastqFile:
    """include/dsrc/FastqFile.h, src/FastqFile.cpp: strings up to a newline; an empty string ends the file."""

    def __init__(self):
        self._f = None
        self._writing = False

    def Open(self, filename: str) -> None:
        if self._f is not None:
            raise RuntimeError("Invalid state")
        self._f = open(filename, "rb"); self._writing = False

    def Create(self, filename: str) -> None:
        if self._f is not None:
            raise RuntimeError("Invalid state")
        self._f = open(filename, "wb"); self._writing = True

    def Close(self) -> None:
        if self._f is None:
            raise RuntimeError("Invalid state")
        self._f.close(); self._f = None

    def ReadNextRecord(self, rec: FastqRecord) -> bool:
        if self._f is None or self._writing:
            raise RuntimeError("Invalid state")
        parts = []
        for _ in range(4):
            line = self._f.readline()
            if line.endswith(b"\n"):
                line = line[:-1]
            if not line:
                for name, v in zip(FastqRecord.__slots__, parts + [b""] * (4 - len(parts))):
                    setattr(rec, name, v.decode("latin-1"))
                return False
            parts.append(line)
        rec.tag, rec.sequence, rec.plus, rec.quality = (p.decode("latin-1") for p in parts)
        return True

    def WriteNextRecord(self, rec: FastqRecord) -> None:
        if self._f is None or not self._writing:
            raise RuntimeError("Invalid state")
        self._f.write("\n".join((rec.tag, rec.sequence, rec.plus, rec.quality)).encode("latin-1") + b"\n")


_RECORDS_CLI = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "dsrc-amd-records")


class DsrcArchive:
    """Write side of the record-level archive API (py/Interface.cpp:78-94, src/DsrcArchive.cpp): the records go to the
    C++ host's DsrcArchive (dsrc_amd/csrc/host, `dsrc-amd-records` reading them from a pipe), which cuts chunks like
    BlockCompressorExt and compresses them on the GPU; the archive is the one the reference's DsrcArchive writes.
    As there, QualityCompressionLevel 1-2 is only defined together with LossyCompression, and Crc32Checking /
    TagFieldFilterMask are accepted and ignored.  Reading needs the block decompressor."""

    def __init__(self):
        self.LossyCompression = False
        self.DNACompressionLevel = 0
        self.QualityCompressionLevel = 0
        self.TagFieldFilterMask = 0
        self.PlusRepetition = False
        self.QualityOffset = 0
        self.ColorSpace = False
        self.FastqBufferSizeMB = 8
        self.Crc32Checking = False
        self.Device = 0
        self._proc = None

    def StartCompress(self, filename: str) -> None:
        if self._proc is not None:
            raise RuntimeError("Invalid state")
        if not 0 <= int(self.DNACompressionLevel) <= 3:
            raise RuntimeError("Invalid DNA compression mode specified [0-3]")
        if not 0 <= int(self.QualityCompressionLevel) <= 2:
            raise RuntimeError("Invalid Quality compression mode specified [0-2]")
        if not 1 <= int(self.FastqBufferSizeMB) <= 1024:
            raise RuntimeError("Invalid fastq buffer size specified [1-1024]")
        if self.ColorSpace:
            raise RuntimeError("colour-space records are not supported by the record-level API on the GPU path (use DsrcModule.Compress)")
        if not os.path.exists(_RECORDS_CLI):
            raise RuntimeError(f"{_RECORDS_CLI} not built: python -c 'import __graft_entry__ as g; g.build()'")
        cmd = [_RECORDS_CLI, "/dev/stdin", filename, str(int(self.DNACompressionLevel)), str(int(self.QualityCompressionLevel)),
               "1" if self.LossyCompression else "0", str(int(self.FastqBufferSizeMB)), str(int(self.QualityOffset)),
               "1" if self.PlusRepetition else "0", str(int(self.Device))]
        self._proc = subprocess.Popen(cmd, stdin=subprocess.PIPE, stderr=subprocess.PIPE)

    def _fail(self):
        p, self._proc = self._proc, None
        try:
            p.stdin.close()
        except OSError:
            pass
        err = p.stderr.read().decode(errors="replace").strip()
        p.wait()
        raise RuntimeError(err or "dsrc-amd-records failed")

    def WriteNextRecord(self, rec: FastqRecord) -> None:
        if self._proc is None:
            raise RuntimeError("Invalid state")
        parts = (rec.tag, rec.sequence, rec.plus, rec.quality)
        if (not rec.tag.startswith("@") or not rec.plus.startswith("+") or not rec.sequence or len(rec.sequence) != len(rec.quality)
                or any("\n" in p or "\r" in p for p in parts)):
            raise RuntimeError("DsrcArchive.WriteNextRecord: malformed record")
        try:
            self._proc.stdin.write("\n".join(parts).encode("latin-1") + b"\n")
        except BrokenPipeError:
            self._fail()

    def FinishCompress(self) -> None:
        if self._proc is None:
            raise RuntimeError("Invalid state")
        p = self._proc
        try:
            p.stdin.close()
        except BrokenPipeError:
            pass
        err = p.stderr.read().decode(errors="replace").strip()
        rc = p.wait()
        self._proc = None
        if rc != 0:
            raise RuntimeError(err or "dsrc-amd-records failed")

    def StartDecompress(self, filename: str) -> None:
        raise RuntimeError("DsrcArchive: reading records needs the block decompressor, which is not part of the MI355X path yet (SURVEY 8f-1)")

    def ReadNextRecord(self, rec: FastqRecord) -> bool:
        raise RuntimeError("Invalid state")

    def FinishDecompress(self) -> None:
        raise RuntimeError("Invalid state")
