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


def test_record_api_names(tmp_path):
    a = pydsrc.DsrcArchive()
    for name in ("StartCompress", "WriteNextRecord", "FinishCompress", "StartDecompress", "ReadNextRecord", "FinishDecompress",
                 "LossyCompression", "DNACompressionLevel", "QualityCompressionLevel", "TagFieldFilterMask", "PlusRepetition",
                 "QualityOffset", "ColorSpace", "FastqBufferSizeMB", "Crc32Checking"):
        assert hasattr(a, name)
    with pytest.raises(RuntimeError):
        a.StartDecompress("x.dsrc")
    with pytest.raises(RuntimeError):
        a.WriteNextRecord(pydsrc.FastqRecord())          # not started
    assert pydsrc.FieldMask().AddField(1).AddField(2).GetMask() == 6
    # FastqFile: round trip, an empty line ends the file (src/FastqFile.cpp:66-92)
    p = str(tmp_path / "a.fastq")
    f = pydsrc.FastqFile(); f.Create(p)
    r = pydsrc.FastqRecord(); r.tag, r.sequence, r.plus, r.quality = "@a 1", "ACGT", "+", "IIII"
    f.WriteNextRecord(r); r.tag = "@a 2"; f.WriteNextRecord(r); f.Close()
    assert open(p, "rb").read() == b"@a 1\nACGT\n+\nIIII\n@a 2\nACGT\n+\nIIII\n"
    f = pydsrc.FastqFile(); f.Open(p)
    q = pydsrc.FastqRecord(); tags = []
    while f.ReadNextRecord(q):
        tags.append(q.tag)
    f.Close()
    assert tags == ["@a 1", "@a 2"]
    with pytest.raises(RuntimeError):
        f.Close()
