"""`pydsrc` on the MI355X path (reference: py/Interface.cpp:55-109, boost::python over wrap::DsrcModule / DsrcArchive /
FastqFile).  Same class, method and property names, so a script written for the reference's module runs unchanged:

    from dsrc_amd import pydsrc
    m = pydsrc.DsrcModule()
    m.DNACompressionLevel = 3; m.QualityCompressionLevel = 2
    m.Compress("in.fastq", "out.dsrc")          # the archive `dsrc c -t1` writes
    m.Decompress("out.dsrc", "back.fastq")

The classes are a pybind11 extension (dsrc_amd/_pydsrc, source dsrc_amd/csrc/host/pydsrc_module.cpp) bound in-process to the
C++ host classes of dsrc_amd/csrc/host/dsrc_host.h, which drive the GPU through the C ABI (include/dsrc_gpu.h).  Any
DsrcException surfaces as RuntimeError, as with the reference's exception translator.  There is no CPU codec behind it:
without the built extension the import fails, without a GPU the calls that need one raise RuntimeError.

Note: the reference binds the *setter* of QualityCompressionLevel to SetDnaCompressionLevel (py/Interface.cpp:88,103), so
assigning it there silently changes the DNA level; here the property sets the quality level.  `Device` (GPU ordinal) is
an extra property.
"""
from __future__ import annotations

try:
    from dsrc_amd._pydsrc import DsrcArchive, DsrcModule, FastqFile, FastqRecord, FieldMask  # noqa: F401
except ImportError as e:  # pragma: no cover
    raise ImportError("dsrc_amd._pydsrc is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(pybind11 over dsrc_amd/csrc/host; needs libdsrc_gpu.so). There is no pure-Python fallback.") from e

__all__ = ["DsrcArchive", "DsrcModule", "FastqFile", "FastqRecord", "FieldMask"]
