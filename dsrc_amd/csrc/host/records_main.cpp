// dsrc-amd-records: the reference's examples/cpplib/example2.cpp use case on the MI355X path -- read a FASTQ file record
// by record (FastqFile) and write it through the record-level archive API (DsrcArchive::WriteNextRecord).
//   dsrc-amd-records <in.fastq> <out.dsrc> <dnaLevel> <qualityLevel> <lossy 0|1> <bufferMB> <qualityOffset> [plusRepetition] [device]
// and the way back (examples/cpplib/example3.cpp): DsrcArchive::ReadNextRecord -> FastqFile::WriteNextRecord
//   dsrc-amd-records -x <in.dsrc> <out.fastq> [device]
#include <cstdio>
#include <cstdlib>
#include <string>

#include "dsrc_host.h"

int main(int argc, char** argv)
{
	if (argc >= 4 && std::string(argv[1]) == "-x")
	{
		using namespace dsrc::wrap;
		try
		{
			DsrcArchive ar;
			if (argc > 4) ar.SetDevice(std::atoi(argv[4]));
			ar.StartDecompress(argv[2]);
			FastqFile out;
			out.Create(argv[3]);
			FastqRecord rec;
			unsigned long n = 0;
			while (ar.ReadNextRecord(rec)) { out.WriteNextRecord(rec); n++; }
			ar.FinishDecompress();
			out.Close();
			std::fprintf(stderr, "records: %lu (offset %u, dna %u, quality %u, lossy %d)\n", n, ar.GetQualityOffset(), ar.GetDnaCompressionLevel(),
						 ar.GetQualityCompressionLevel(), (int)ar.IsLossyCompression());
		}
		catch (const dsrc::DsrcException& e) { std::fprintf(stderr, "Error: %s\n", e.what()); return 1; }
		return 0;
	}
	if (argc < 8) { std::fprintf(stderr, "usage: dsrc-amd-records in.fastq out.dsrc dna qua lossy bufMB offset [plusrep] [device]\n"); return 2; }
	using namespace dsrc::wrap;
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
		if (argc > 9) ar.SetDevice(std::atoi(argv[9]));
		ar.StartCompress(argv[2]);
		FastqRecord rec;
		unsigned long n = 0;
		while (in.ReadNextRecord(rec)) { ar.WriteNextRecord(rec); n++; }
		ar.FinishCompress();
		in.Close();
		std::fprintf(stderr, "records: %lu\n", n);
	}
	catch (const dsrc::DsrcException& e) { std::fprintf(stderr, "Error: %s\n", e.what()); return 1; }
	return 0;
}
