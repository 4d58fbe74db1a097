"""`pydsrc`-compatible compression front end (reference: py/Interface.cpp:55-109, boost::python over
DsrcModule).  Same class and property names, so a script written for the reference's module runs unchanged for
compression:

    from dsrc_amd import pydsrc
    m = pydsrc.DsrcModule()
    m.DNACompressionLevel = 3; m.QualityCompressionLevel = 2
    m.Compress("in.fastq", "out.dsrc")

It drives the C++ host pipeline (`dsrc-amd`, dsrc_amd/csrc/host) on top of the C ABI; the archive is the one
`dsrc c -t1` writes.  Decompression and the record-level DsrcArchive API are not part of the MI355X path
(SURVEY 8f-1 / 8f-3): they raise RuntimeError, like the reference's module does for any DsrcException.

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


class DsrcArchive:
    def __init__(self):
        raise RuntimeError("The record-level DsrcArchive API is not part of the MI355X hot path (SURVEY 8f-3)")
