// dsrc-amd: command line of the MI355X compressor.  Same switches as the reference's `dsrc c`
// (src/main.cpp:137-308): -d<n> -q<n> -l -c -o<n> -b<n> -m<n> -v [-t<n> accepted and ignored], plus -g<dev>, -n<blocks/batch>.
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "dsrc_host.h"

using namespace dsrc;

int main(int argc, const char* argv[])
{
	setenv("GPU_MAX_HW_QUEUES", "24", 0);      // two HIP streams per scheduler instance; before the runtime starts
	if (argc < 4 || argv[1][0] != 'c')
	{
		std::cerr << "usage: dsrc-amd c [-d<0-3>] [-q<0-2>] [-l] [-c] [-f<fields>] [-o<offset>] [-b<MB>] [-m<0-2>] [-v] [-g<device>] [-n<blocks per batch>] <in.fastq> <out.dsrc>\n"
					 "       (decompression: use the reference `dsrc d`; archives are bit-identical)\n";
		return -1;
	}
	comp::InputParameters p;
	bool verbose = false;
	for (int i = 2; i < argc - 2; ++i)
	{
		const char* a = argv[i];
		if (a[0] != '-') continue;
		const int v = strlen(a) > 2 ? atoi(a + 2) : -1;
		switch (a[1])
		{
		case 'f':       // -f<1,2,...>: keep only these title fields (reference src/main.cpp:177-194)
		{
			const char* q = a + 2;
			while (*q)
			{
				const int f = atoi(q);
				if (f >= 1 && f <= 30) p.tagPreserveFlags |= 1ull << f;
				else { std::cerr << "Error: invalid field number in -f (1-30)\n"; return -1; }
				while (*q && *q != ',') ++q;
				if (*q == ',') ++q;
			}
			break;
		}
		case 'o': p.qualityOffset = v; break;
		case 'd': p.dnaCompressionLevel = v; break;
		case 'q': p.qualityCompressionLevel = v; break;
		case 't': p.threadNum = v; break;
		case 'b': p.fastqBufferSizeMB = v; break;
		case 'l': p.lossyCompression = true; break;
		case 'c': p.calculateCrc32 = true; break;
		case 'v': verbose = true; break;
		case 'g': p.device = v; break;
		case 'n': p.batchBlocks = v; break;
		case 'm':
			if (v == 2) { p.dnaCompressionLevel = 3; p.qualityCompressionLevel = 2; p.fastqBufferSizeMB = 256; }
			else if (v == 1) { p.dnaCompressionLevel = 2; p.qualityCompressionLevel = 2; p.fastqBufferSizeMB = 64; }
			else if (v == 0) { p.dnaCompressionLevel = 0; p.qualityCompressionLevel = 0; p.fastqBufferSizeMB = 8; }
			break;
		}
	}
	p.inputFilename = argv[argc - 2]; p.outputFilename = argv[argc - 1];
	if (p.dnaCompressionLevel > 3 || p.qualityCompressionLevel > 2 || p.fastqBufferSizeMB < 1 || p.fastqBufferSizeMB > 1024)
	{
		std::cerr << "Error: invalid compression parameters\n";
		return -1;
	}
	comp::DsrcCompressorGPU op;
	if (!op.Process(p)) { std::cerr << op.GetError(); return -1; }
	if (verbose) std::cout << op.GetLog();
	return 0;
}
