"""CPU: the pydsrc-compatible front end keeps the reference's names and range checks (py/Interface.cpp:96-108)."""
import pytest

from dsrc_amd import pydsrc


def test_properties_and_checks():
    m = pydsrc.DsrcModule()
    for name in ("LossyCompression", "DNACompressionLevel", "QualityCompressionLevel", "TagFieldFilterMask",
                 "FastqBufferSizeMB", "ThreadsNumber", "Crc32Checking", "Compress", "Decompress"):
        assert hasattr(m, name)
    m.DNACompressionLevel = 3; m.QualityCompressionLevel = 2
    assert (m.DNACompressionLevel, m.QualityCompressionLevel) == (3, 2)      # the reference's setter bug is not reproduced
    for attr, bad in (("DNACompressionLevel", 4), ("QualityCompressionLevel", 3), ("FastqBufferSizeMB", 0), ("ThreadsNumber", 65)):
        with pytest.raises(RuntimeError):
            setattr(m, attr, bad)
    with pytest.raises(RuntimeError):
        pydsrc.DsrcArchive()
