// dsrc-amd: command line of the MI355X DSRC path.  Same modes and switches as the reference's `dsrc`
// (src/main.cpp:94-308): `c` with -d<n> -q<n> -l -c -f<..> -o<n> -b<n> -m<n>, `d`, and for both -t<n> -s -v.
// Here -t<n> is the number of GPU scheduler instances (host threads working on consecutive batches, 1..8 are used;
// the range check is the reference's 1..64) PER DEVICE; extra switches: -g<device>[,<device>..] (GPUs of the node to use), -n<blocks per batch>,
// -x (with -c: store the checksums but skip the decode-and-compare pass the reference runs after every block).
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <unistd.h>

#include "dsrc_host.h"

using namespace dsrc;

static int usage()
{
	std::cerr << "usage: dsrc-amd <c|d> [options] <input filename> <output filename>\n"
				 "compression options: -d<0-3> -q<0-2> -f<1,..> -b<MB> -o<offset> -l -c -m<0-2>\n"
				 "both: -t<n> (GPU scheduler instances) -s (stdin/stdout for raw FASTQ) -v   GPU: -g<device>[,<device>..] -n<blocks per batch> -x (no verify pass with -c)\n";
	return -1;
}

int main(int argc, const char* argv[])
{
	setenv("GPU_MAX_HW_QUEUES", "24", 0);      // two HIP streams per scheduler instance; before the runtime starts
	if (argc < 3 || (argv[1][0] != 'c' && argv[1][0] != 'd') || argv[1][1] != 0) return usage();
	const bool compress = argv[1][0] == 'c';
	comp::InputParameters p;
	bool verbose = false;
	int nFiles = 0; const char* files[2] = {nullptr, nullptr};
	for (int i = 2; i < argc; ++i)
	{
		const char* a = argv[i];
		if (a[0] != '-')
		{
			if (nFiles == 2) { std::cerr << "Error: too many file names\n"; return -1; }
			files[nFiles++] = a;
			continue;
		}
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
		case 'o': p.qualityOffset = (uint32)v; break;
		case 'd': p.dnaCompressionLevel = (uint32)v; break;
		case 'q': p.qualityCompressionLevel = (uint32)v; break;
		case 't': p.threadNum = (uint32)v; break;
		case 'b': p.fastqBufferSizeMB = (uint32)v; break;
		case 'l': p.lossyCompression = true; break;
		case 'c': p.calculateCrc32 = true; break;
		case 'x': p.verifyCrc32 = false; break;
		case 's': p.useFastqStdIo = true; break;
		case 'v': verbose = true; break;
		case 'g':       // -g<dev>[,<dev>..]: GPUs of this node to spread the scheduler instances over
		{
			const char* q = a + 2;
			p.devices.clear();
			while (*q) { p.devices.push_back(atoi(q)); while (*q && *q != ',') ++q; if (*q == ',') ++q; }
			if (p.devices.empty()) { std::cerr << "Error: -g needs a device number\n"; return -1; }
			p.device = p.devices[0];
			break;
		}
		case 'n': p.batchBlocks = (uint32)v; break;
		case 'm':
			if (v == 2) { p.dnaCompressionLevel = 3; p.qualityCompressionLevel = 2; p.fastqBufferSizeMB = 256; }
			else if (v == 1) { p.dnaCompressionLevel = 2; p.qualityCompressionLevel = 2; p.fastqBufferSizeMB = 64; }
			else if (v == 0) { p.dnaCompressionLevel = 0; p.qualityCompressionLevel = 0; p.fastqBufferSizeMB = 8; }
			else { std::cerr << "Error: invalid compression mode specified [0-2]\n"; return -1; }
			break;
		default:
			std::cerr << "Error: unknown option " << a << '\n';
			return -1;
		}
	}
	// file names as in the reference (src/main.cpp:225-241): with -s the FASTQ side is stdin / stdout
	if (nFiles != (p.useFastqStdIo ? 1 : 2)) return usage();
	if (!p.useFastqStdIo) { p.inputFilename = files[0]; p.outputFilename = files[1]; }
	else if (compress) p.outputFilename = files[0];
	else p.inputFilename = files[0];
	if (p.inputFilename == p.outputFilename) { std::cerr << "Error: input and output filenames are the same\n"; return -1; }
	// range checks of the reference (src/main.cpp:276-305)
	if (p.qualityOffset != 0 && !(p.qualityOffset >= 33 && p.qualityOffset <= 64)) { std::cerr << "Error: invalid Quality offset mode specified [33, 64]\n"; return -1; }
	if (p.dnaCompressionLevel > 3) { std::cerr << "Error: invalid DNA compression mode specified [0-3]\n"; return -1; }
	if (p.qualityCompressionLevel > 2) { std::cerr << "Error: invalid Quality compression mode specified [0-2]\n"; return -1; }
	if (p.threadNum == 0 || p.threadNum > 64) { std::cerr << "Error: invalid thread number specified [1-64]\n"; return -1; }
	if (!(p.fastqBufferSizeMB >= 1 && p.fastqBufferSizeMB <= 1024)) { std::cerr << "Error: invalid fastq buffer size specified [1-1024] \n"; return -1; }

	p.exitWhenDone = !p.useFastqStdIo; p.verboseLog = verbose;      // a file-to-file run ends when the output is complete
	comp::IDsrcOperator* op = compress ? (comp::IDsrcOperator*)new comp::DsrcCompressorGPU() : (comp::IDsrcOperator*)new comp::DsrcDecompressorGPU();
	const bool ok = op->Process(p);
	if (!ok) std::cerr << op->GetError();
	else if (verbose) std::cerr << op->GetLog();
	delete op;
	if (ok)
	{	// everything is written and closed: leave without the HIP runtime's tear-down (0.3-0.5 s of unmapping at exit)
		std::cerr.flush(); fflush(nullptr);
		_exit(0);
	}
	return -1;
}
