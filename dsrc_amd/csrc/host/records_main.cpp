// dsrc-amd-records: the reference's examples/cpplib/example2.cpp use case on the MI355X path -- read a FASTQ file record
// by record (FastqFile) and write it through the record-level archive API (DsrcArchive::WriteNextRecord).
//   dsrc-amd-records <in.fastq> <out.dsrc> <dnaLevel> <qualityLevel> <lossy 0|1> <bufferMB> <qualityOffset> [plusRepetition] [device]
#include <cstdio>
#include <cstdlib>

#include "dsrc_host.h"

int main(int argc, char** argv)
{
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
