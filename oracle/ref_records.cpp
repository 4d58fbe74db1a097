// TEST INFRASTRUCTURE ONLY.  Driver of the UNMODIFIED reference's record-level API (include/dsrc/DsrcArchive.h,
// FastqFile.h): reads a FASTQ file record by record and writes it through DsrcArchive::WriteNextRecord, so that the
// archives of dsrc_amd's DsrcArchive can be compared byte for byte.  Our own code; links the reference objects.
//   ref_records <in.fastq> <out.dsrc> <dnaLevel> <qualityLevel> <lossy 0|1> <bufferMB> <qualityOffset> [plusRepetition]
//   ref_records -x <in.dsrc> <out.fastq>      the way back: DsrcArchive::StartDecompress / ReadNextRecord -> FastqFile
#include <cstdio>
#include <cstdlib>
#include "dsrc/Dsrc.h"

int main(int argc, char** argv)
{
	using namespace dsrc::lib;
	if (argc == 4 && argv[1][0] == '-' && argv[1][1] == 'x')
	{
		try
		{
			DsrcArchive ar;
			ar.StartDecompress(argv[2]);
			FastqFile out;
			out.Create(argv[3]);
			FastqRecord rec;
			unsigned long n = 0;
			while (ar.ReadNextRecord(rec)) { out.WriteNextRecord(rec); n++; }
			ar.FinishDecompress();
			out.Close();
			std::fprintf(stderr, "records: %lu\n", n);
		}
		catch (const DsrcException& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
		return 0;
	}
	if (argc < 8) { std::fprintf(stderr, "usage: ref_records in out dna qua lossy bufMB offset [plusrep] | -x in.dsrc out.fastq\n"); return 2; }
	try
	{
		FastqFile in;
		in.Open(argv[1]);
		DsrcArchive ar;
		ar.SetDnaCompressionLevel(std::atoi(argv[3]));
		ar.SetQualityCompressionLevel(std::atoi(argv[4]));
		ar.SetLossyCompression(std::atoi(argv[5]) != 0);
		ar.SetFastqBufferSizeMB(std::atoi(argv[6]));
		ar.SetQualityOffset(std::atoi(argv[7]));
		ar.SetPlusRepetition(argc > 8 && std::atoi(argv[8]) != 0);
		ar.StartCompress(argv[2]);
		FastqRecord rec;
		unsigned long n = 0;
		while (in.ReadNextRecord(rec)) { ar.WriteNextRecord(rec); n++; }
		ar.FinishCompress();
		in.Close();
		std::fprintf(stderr, "records: %lu\n", n);
	}
	catch (const DsrcException& e) { std::fprintf(stderr, "error: %s\n", e.what()); return 1; }
	return 0;
}
