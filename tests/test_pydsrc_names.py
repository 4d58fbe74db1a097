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
    # the setters' ranges and texts are the reference's (src/Configurable.cpp:56-144), not the command line's (src/main.cpp:276-297)
    for attr, bad, text in (("DNACompressionLevel", 4, "Invalid argument: invalid DNA compression level [0-3]"),
                            ("QualityCompressionLevel", 3, "Invalid argument: invalid Quality compression level [0-2]"),
                            ("FastqBufferSizeMB", 0, "Invalid argument: invalid FASTQ buffer size [1-1024]"),
                            ("FastqBufferSizeMB", 1025, "Invalid argument: invalid FASTQ buffer size [1-1024]"),
                            ("ThreadsNumber", 0, "Invalid argument: thread number must be greater than 0")):
        with pytest.raises(RuntimeError) as e:
            setattr(m, attr, bad)
        assert text in str(e.value)
    m.ThreadsNumber = 65; m.FastqBufferSizeMB = 1024                         # accepted there, accepted here
    assert (m.ThreadsNumber, m.FastqBufferSizeMB) == (65, 1024)
    a = pydsrc.DsrcArchive()
    for bad in (0, 34, 63, 65):
        with pytest.raises(RuntimeError) as e:
            a.QualityOffset = bad
        assert "Invalid argument: only valid Quality offset are 33 and 64" in str(e.value)
    a.QualityOffset = 64; a.QualityOffset = 33
    assert a.QualityOffset == 33


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
