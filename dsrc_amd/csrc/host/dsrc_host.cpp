// Host side of the MI355X DSRC compressor (see dsrc_host.h).  Plain C++17, links libdsrc_gpu.so.
#ifndef _GNU_SOURCE
#define _GNU_SOURCE            // O_DIRECT
#endif
#include "dsrc_host.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cerrno>
#include <cstring>
#include <deque>
#include <iomanip>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>
#include <chrono>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "dsrc_gpu.h"

namespace dsrc
{
namespace comp
{

CompressionSettings IDsrcOperator::GetCompressionSettings(const InputParameters& a)
{
	CompressionSettings s;
	s.lossy = a.lossyCompression;
	s.dnaOrder = a.dnaCompressionLevel * 3;
	s.qualityOrder = s.lossy ? a.qualityCompressionLevel * 3 : a.qualityCompressionLevel;
	s.tagPreserveFlags = a.tagPreserveFlags;
	s.calculateCrc32 = a.calculateCrc32;
	return s;
}

// ---- chunk cutter ---------------------------------------------------------------------------------------
FastqChunker::FastqChunker(FILE* f, uint64 bufferSize) : file(f), bufSize(bufferSize) {}

static void SkipToEol(const uchar* d, uint64& pos, uint64 size, bool& crlf)
{
	while (pos < size && d[pos] != '\n' && d[pos] != '\r') ++pos;
	if (pos < size && d[pos] == '\r' && pos + 1 < size && d[pos + 1] == '\n') { crlf = true; ++pos; }
}

// first record start after `pos`: a line beginning with '@' whose successor is not itself an '@' line
// (a quality line may start with '@'), src/FastqStream.cpp:74-98
uint64 FastqChunker::NextRecordPos(const uchar* d, uint64 pos, uint64 size, bool& crlf)
{
	SkipToEol(d, pos, size, crlf); ++pos;
	while (pos < size && d[pos] != '@') { SkipToEol(d, pos, size, crlf); ++pos; }
	const uint64 candidate = pos;
	SkipToEol(d, pos, size, crlf); ++pos;
	if (pos < size && d[pos] == '@') return pos;
	return candidate;
}

bool FastqChunker::ReadNextChunk(std::vector<uchar>& chunk)
{
	if (eof) { chunk.clear(); return false; }
	chunk.resize(bufSize);
	uint64 have = carry.size();
	if (have) std::copy(carry.begin(), carry.end(), chunk.begin());
	carry.clear();
	const uint64 toRead = bufSize - have;
	const uint64 r = fread(chunk.data() + have, 1, toRead, file);
	if (r == 0) { eof = true; chunk.resize(have); return true; }
	if (r == toRead)
	{
		const uint64 end = NextRecordPos(chunk.data(), bufSize - 8192, bufSize, usesCrlf);
		if (end > bufSize) throw DsrcException("no record boundary in the last 8 KiB of the FASTQ buffer (a record longer than 8 KiB is undefined in the reference's reader, src/FastqStream.cpp:74-98); use a larger -b");
		carry.assign(chunk.begin() + end, chunk.end());
		chunk.resize(end - 1 - (usesCrlf ? 1 : 0));
		return true;
	}
	eof = true;
	chunk.resize(have + r - 1 - (usesCrlf ? 1 : 0));
	return true;
}

// ---- first-chunk analysis ---------------------------------------------------------------------------------
namespace
{
struct LineScanner
{
	const uchar* d; uint64 size, pos;
	uint32 Skip()          // one line, "\r\n" / "\n" / "\r" terminated (src/FastqParser.h:93-115)
	{
		uint32 len = 0;
		while (pos < size)
		{
			const uchar c = d[pos++];
			if (c != '\n' && c != '\r') { ++len; continue; }
			if (c == '\r' && pos < size && d[pos] == '\n') ++pos;
			break;
		}
		return len;
	}
};
}

bool AnalyzeFirstChunk(const uchar* data, uint64 size, fq::FastqDatasetType& type, bool estimate)
{
	LineScanner s{data, size, 0};
	uchar minQ = 255, maxQ = 0;
	type.colorSpace = false; type.plusRepetition = false;
	uint32 count = 0;
	while (s.pos < s.size)
	{
		const uchar* title = data + s.pos; if (s.Skip() == 0 || title[0] != '@') break;
		const uchar* seq = data + s.pos;   if (s.Skip() == 0) break;
		const uchar* plus = data + s.pos;  const bool rep = s.Skip() > 1; if (plus[0] != '+') break;
		const uchar* qua = data + s.pos;   const uint32 ql = s.Skip();
		if (estimate) for (uint32 i = 0; i < ql; ++i) { minQ = std::min(minQ, qua[i]); maxQ = std::max(maxQ, qua[i]); }
		else if (ql == 0) break;
		const bool cs = (seq[1] >= '0' && seq[1] <= '3') || seq[1] == '.';
		if (count)
		{
			if (type.colorSpace != cs) return false;
			if (cs && seq[0] >= '0' && seq[0] <= '3') return false;
			if (type.plusRepetition != rep) return false;
		}
		else { type.plusRepetition = rep; type.colorSpace = cs; }
		++count;
	}
	if (estimate)
	{
		if (maxQ <= 74) { if (minQ >= 33) type.qualityOffset = 33; }
		else if (maxQ <= 105) { if (minQ >= 64) type.qualityOffset = 64; else if (minQ >= 59) type.qualityOffset = 59; }
		if (type.qualityOffset == 0) { if (minQ >= 33) type.qualityOffset = 33; else return false; }
	}
	return count > 1;
}

// ---- archive ------------------------------------------------------------------------------------------------
static void PutBE(std::vector<uchar>& v, uint64 x, int bytes) { for (int i = bytes - 1; i >= 0; --i) v.push_back((uchar)(x >> (8 * i))); }

void ArchiveWriter::Start(const std::string& path)
{
	fd = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
	if (fd < 0) throw DsrcException("Cannot open file to write:" + path);
	name = path;
	pos = 40;                                        // the header is written last (src/DsrcFile.cpp:52-54)
}

void ArchiveWriter::WriteAt(uint64 off, const void* p, uint64 n) const
{
	const uchar* b = (const uchar*)p;
	while (n)
	{
		const ssize_t w = pwrite(fd, b, n > (1ull << 30) ? (1ull << 30) : n, (off_t)off);
		if (w <= 0) throw DsrcException("Error writing the archive (disk full?): " + name);
		b += w; off += (uint64)w; n -= (uint64)w;
	}
}

void ArchiveWriter::ReserveAhead(uint64 expect)
{
	struct stat sb;
	if (fd < 0 || reserver.joinable() || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) return;
	reserver = std::thread([this, expect]()
	{
		const uint64 piece = 256ull << 20;
		for (uint64 o = 0; o < expect && !reserveStop.load(); o += piece)
		{
			if (fallocate(fd, 0, (off_t)o, (off_t)std::min(piece, expect - o)) != 0) return;       // not supported / no space: the writes report it
			reserved.store(o + std::min(piece, expect - o));
		}
	});
}

void ArchiveWriter::StopReserver()
{
	reserveStop.store(true);
	if (reserver.joinable()) reserver.join();
}

void ArchiveWriter::Abandon()
{
	StopReserver();
	if (fd >= 0) { close(fd); fd = -1; }
	if (!name.empty()) { unlink(name.c_str()); name.clear(); }
}

// the next n blocks of the archive: their sizes go into the footer table, the bytes may be written by the caller (any
// thread, any time before Finish) at the returned file offset
uint64 ArchiveWriter::Claim(uint32 n, const uint64_t* sizes, const uint64_t* raw, const uint64_t* comp)
{
	const uint64 at = pos;
	for (uint32 i = 0; i < n; ++i)
	{
		blockSizes.push_back((uint32)sizes[i]); pos += sizes[i];
		for (int k = 0; k < 4; ++k) { rawInfo.sizes[k] += raw[4 * i + k]; compInfo.sizes[k] += comp[4 * i + k]; }
	}
	return at;
}

void ArchiveWriter::WriteBlock(const uchar* data, uint64 size, const uint64 raw[4], const uint64 comp[4])
{
	const uint64_t sz = size, r[4] = {raw[0], raw[1], raw[2], raw[3]}, c[4] = {comp[0], comp[1], comp[2], comp[3]};
	WriteAt(Claim(1, &sz, r, c), data, size);
}

void ArchiveWriter::Finish(const fq::FastqDatasetType& type, const CompressionSettings& s)
{
	const uint64 footerOffset = pos;
	std::vector<uchar> foot;
	foot.push_back(0xCC);
	const uchar* bs = (const uchar*)blockSizes.data();     // host-endian uint32 array, as the reference writes it
	foot.insert(foot.end(), bs, bs + blockSizes.size() * 4);
	foot.push_back((uchar)((type.colorSpace ? 2 : 0) | (type.plusRepetition ? 1 : 0)));
	foot.push_back((uchar)type.qualityOffset);
	foot.push_back((uchar)((s.lossy ? 1 : 0) | (s.calculateCrc32 ? 2 : 0)));
	foot.push_back((uchar)s.dnaOrder); foot.push_back((uchar)s.qualityOrder);
	PutBE(foot, s.tagPreserveFlags, 8);
	WriteAt(footerOffset, foot.data(), foot.size());
	std::vector<uchar> head;
	head.push_back(0xAA); head.push_back(2); head.push_back(0); head.push_back(2);
	PutBE(head, foot.size(), 4); PutBE(head, footerOffset, 8); PutBE(head, 0, 8); PutBE(head, blockSizes.size(), 8);
	for (int i = 0; i < 8; ++i) head.push_back(0xAA);
	WriteAt(0, head.data(), head.size());
	StopReserver();
	if (reserved.load() > footerOffset + foot.size() && ftruncate(fd, (off_t)(footerOffset + foot.size())) != 0)
		throw DsrcException("Error writing the archive: " + name);
	const int g = fd; fd = -1;
	if (close(g) != 0) throw DsrcException("Error writing the archive (disk full?): " + name);
	name.clear();
}

ArchiveWriter::~ArchiveWriter() { if (fd >= 0) Abandon(); }           // an archive that was not finished is not left behind

// ---- operator -----------------------------------------------------------------------------------------------
// File -> archive as a pipeline: the calling thread cuts chunk boundaries (two 8 KiB reads per chunk),
// reader threads pread() whole batches straight into page-locked memory, `instances` worker threads each own one GPU
// scheduler instance, compress a batch (host->device, kernels, device->host all on that instance's stream, overlapping
// the other instances) and write its blocks themselves with one positioned write: a batch claims its range of the archive
// as soon as the batches before it have claimed theirs (that needs their sizes, not their bytes), so writes run in parallel.  The block-to-block state
// travels through a dsrcgpu_chain, so the archive is the one a single instance -- and `dsrc c -t1` -- writes.
namespace
{
struct Pinned
{
	uchar* p = nullptr; uint64 cap = 0; bool pageable = false;
	void Reserve(uint64 n)
	{
		if (n <= cap) return;
		Release();
		// Plain (pageable) memory by default: page-locking costs ~0.23 s per 1.6 GB and the runtime serialises it with every
		// other HIP call of the process, so locking the ~13 GB of batch buffers held the first passes back by 1.5-2 s
		// (measured: 38.5 GB file 6.7 s -> 4.8 s); the copies are staged by the runtime, inside the worker's own thread.
		// DSRC_HOST_PINNED=1 brings the page-locked buffers back.
		static const bool usePageable = getenv("DSRC_HOST_PINNED") == nullptr;
		void* q = nullptr;
		if (usePageable)
		{
			if (posix_memalign(&q, 2u << 20, n) != 0) throw DsrcException("out of memory");
			pageable = true;
#ifdef MADV_HUGEPAGE
			// 2 MiB pages where the kernel hands them out on request: a 1.6 GB batch buffer is 800 page faults instead of 400 k when it is
			// first filled, the runtime pins 800 pages for the copy, and the process frees 800 when it leaves (six such buffers of
			// 4 KiB pages: 0.7 s between "archive closed" and the end of the process, profiles/r05_e2e_second.txt)
			(void)madvise(q, n, MADV_HUGEPAGE);
#endif
		}
		else if (dsrcgpu_host_alloc(n, &q) != DSRCGPU_OK) throw DsrcException("cannot allocate page-locked host memory");
		p = (uchar*)q; cap = n;
	}
	void Release() { if (p) { if (pageable) free(p); else dsrcgpu_host_free(p); } p = nullptr; cap = 0; pageable = false; }
	~Pinned() { Release(); }
};

struct Job          // one batch
{
	uint64 seq = 0;
	std::vector<uint64> fileOff; std::vector<uint64_t> sizes;          // chunk i = file[fileOff[i], +sizes[i])
	// filled by a reader thread
	Pinned* in = nullptr; std::vector<uint64> at; uint64 inBytes = 0;
	uint32 nextPart = 0, chunksLeft = 0;                               // reading in parts (guarded by Pipeline::m)
	// results
	std::vector<uint64_t> offs, osz, raw, comp;
};

struct Pipeline
{
	std::mutex m; std::condition_variable cv;
	std::deque<Job*> todo; bool noMore = false;          // cut, waiting for a reader
	std::vector<Job*> reading;                           // being read into their buffer, part by part
	std::map<uint64, Job*> ready; uint64 nextStart = 0;  // read into page-locked memory, waiting for a scheduler instance (taken in order)
	std::vector<Pinned*> freeIn;                         // input buffers not in use
	uint64 claimTurn = 0;                                // the batch whose turn it is to claim its range of the archive
	uint64 written = 0;                                  // batches whose blocks are in the archive file
	std::string error;
	bool Failed() { std::lock_guard<std::mutex> g(m); return !error.empty(); }
	void Fail(const std::string& e) { std::lock_guard<std::mutex> g(m); if (error.empty()) error = e; cv.notify_all(); }
};

// chunk boundaries of a regular file, same decisions as FastqChunker::ReadNextChunk
struct FileCutter
{
	int fd; uint64 fileSize, bufSize, pos = 0, bufEnd = 0; bool eof = false, crlf = false, first = true;
	std::vector<uchar> win;
	FileCutter(int fd_, uint64 size_, uint64 buf_) : fd(fd_), fileSize(size_), bufSize(buf_), win(8192) {}
	bool Next(uint64& start, uint64& size)
	{
		if (eof) return false;
		const uint64 rem = fileSize - pos;
		start = pos;
		if (rem >= bufSize)
		{
			if (pread(fd, win.data(), 8192, (off_t)(pos + bufSize - 8192)) != 8192) throw DsrcException("read error");
			// the window is the last 8 KiB of the buffer: positions are relative to it
			const uint64 wend = FastqChunker::NextRecordPos(win.data(), 0, 8192, crlf);
			if (wend > 8192) throw DsrcException("no record boundary in the last 8 KiB of the FASTQ buffer (a record longer than 8 KiB is undefined in the reference's reader, src/FastqStream.cpp:74-98); use a larger -b");
			const uint64 end = bufSize - 8192 + wend;
			size = end - 1 - (crlf ? 1 : 0);
			bufEnd = pos + bufSize; pos += end; first = false;
			return true;
		}
		eof = true;
		const uint64 have = first ? 0 : bufEnd - pos;          // bytes the stream reader would have carried over
		if (rem == have) { size = have; return true; }          // nothing left to read: the carried bytes as they are
		size = rem - 1 - (crlf ? 1 : 0);
		return true;
	}
};
} // namespace

bool DsrcCompressorGPU::ProcessStream(const InputParameters& args, FILE* in)
{	// stdin: one scheduler instance, batches one after the other
	dsrcgpu_handle* h = nullptr;
	try
	{
		const CompressionSettings settings = GetCompressionSettings(args);
		const uint64 bufSize = (uint64)args.fastqBufferSizeMB << 20;
		FastqChunker chunker(in, bufSize);
		ArchiveWriter writer;
		writer.Start(args.outputFilename);
		std::vector<std::vector<uchar>> chunks;
		chunks.emplace_back();
		fq::FastqDatasetType type;
		const bool findOffset = args.qualityOffset == fq::FastqDatasetType::AutoQualityOffset;
		if (!findOffset) type.qualityOffset = args.qualityOffset;
		if (!chunker.ReadNextChunk(chunks[0]) || !AnalyzeFirstChunk(chunks[0].data(), chunks[0].size(), type, findOffset))
			throw DsrcException("Error analyzing FASTQ dataset");
		h = CreateInstance(args, settings, type);
		const uint32 batch = args.batchBlocks ? args.batchBlocks : (uint32)std::max<uint64>(1, (2048ull << 20) / bufSize);
		bool more = true;
		while (more || !chunks.empty())
		{
			while (more && chunks.size() < batch)
			{
				chunks.emplace_back();
				if (!chunker.ReadNextChunk(chunks.back())) { chunks.pop_back(); more = false; }
			}
			if (chunks.empty()) break;
			const uint32 n = (uint32)chunks.size();
			std::vector<const uint8_t*> ptrs(n); std::vector<uint64_t> sizes(n), offs(n), osz(n), raw(4 * n), comp(4 * n);
			uint64 cap = 0;
			for (uint32 i = 0; i < n; ++i) { ptrs[i] = chunks[i].data(); sizes[i] = chunks[i].size(); cap += sizes[i] + (1u << 16); }
			std::vector<uchar> out(cap);
			if (dsrcgpu_compress_batch(h, n, ptrs.data(), sizes.data(), out.data(), cap, offs.data(), osz.data(), raw.data(), comp.data()) != DSRCGPU_OK)
				throw DsrcException(dsrcgpu_last_error(h));
			for (uint32 i = 0; i < n; ++i) writer.WriteBlock(out.data() + offs[i], osz[i], (const uint64*)&raw[4 * i], (const uint64*)&comp[4 * i]);
			chunks.clear();
		}
		writer.Finish(type, settings);
		LogSizes(writer);
	}
	catch (const DsrcException& e) { AddError(e.what()); }
	catch (const std::exception& e) { AddError(e.what()); }
	if (h) dsrcgpu_destroy(h);
	return !IsError();
}


dsrcgpu_handle* DsrcCompressorGPU::CreateInstance(const InputParameters& args, const CompressionSettings& settings, const fq::FastqDatasetType& type, int device)
{
	dsrcgpu_settings gs; memset(&gs, 0, sizeof(gs));
	gs.dna_order = settings.dnaOrder; gs.quality_order = settings.qualityOrder; gs.tag_preserve_flags = settings.tagPreserveFlags;
	gs.lossy = settings.lossy; gs.calculate_crc32 = settings.calculateCrc32;
	gs.verify_after_compress = settings.calculateCrc32 && args.verifyCrc32;     // what the reference's worker does with -c (src/DsrcWorker.cpp:53-62)
	dsrcgpu_dataset gd; memset(&gd, 0, sizeof(gd));
	gd.quality_offset = type.qualityOffset; gd.plus_repetition = type.plusRepetition; gd.color_space = type.colorSpace;
	dsrcgpu_handle* h = nullptr;
	if (dsrcgpu_create(&gs, &gd, device >= 0 ? device : args.device, 0, &h) != DSRCGPU_OK)
	{
		const std::string msg = h ? dsrcgpu_last_error(h) : "cannot create the GPU compressor";
		if (h) dsrcgpu_destroy(h);
		throw DsrcException(msg);
	}
	return h;
}

void DsrcCompressorGPU::LogSizes(const ArchiveWriter& writer)
{
	std::ostringstream ss;          // same text as the reference's -v log (src/DsrcOperator.cpp:362-375)
	const fq::StreamsInfo& rawS = writer.Raw(); const fq::StreamsInfo& compS = writer.Comp();
	ss << "Compressed streams sizes (in bytes)\n";
	ss << "TAG: " << std::setw(16) << compS.sizes[fq::StreamsInfo::MetaStream] + compS.sizes[fq::StreamsInfo::TagStream]
	   << " / " << std::setw(16) << rawS.sizes[fq::StreamsInfo::TagStream] << '\n';
	ss << "DNA: " << std::setw(16) << compS.sizes[fq::StreamsInfo::DnaStream] << " / " << std::setw(16) << rawS.sizes[fq::StreamsInfo::DnaStream] << '\n';
	ss << "QUA: " << std::setw(16) << compS.sizes[fq::StreamsInfo::QualityStream] << " / " << std::setw(16) << rawS.sizes[fq::StreamsInfo::QualityStream] << '\n';
	AddLog(ss.str());
}

bool DsrcCompressorGPU::Process(const InputParameters& args)
{
	if (args.useFastqStdIo) return ProcessStream(args, stdin);
	int fd = -1, fdDirect = -1;
	dsrcgpu_chain* chain = nullptr;
	std::vector<std::thread> workers, readers;
	Pipeline pl;
	// everything the threads refer to outlives them (they are joined at the bottom, also on errors)
	fq::FastqDatasetType type;
	CompressionSettings settings;
	ArchiveWriter writer;
	std::vector<std::unique_ptr<Pinned>> inBufs;
	uint64 totalBatches = ~0ull;                     // known once the cutter is through (guarded by pl.m)
	const auto tStart = std::chrono::steady_clock::now();
	std::vector<std::thread> warm;                   // HIP start-up runs beside the file opening / first-chunk analysis
	try
	{
		{
			const std::vector<int> d0 = args.devices.empty() ? std::vector<int>(1, args.device) : args.devices;
			for (int dev : d0) warm.emplace_back([dev]() { (void)dsrcgpu_prepare(dev); });
		}
		fd = open(args.inputFilename.c_str(), O_RDONLY);
		struct stat sb;
		if (fd < 0 || fstat(fd, &sb) != 0) throw DsrcException("Cannot open file to read:" + args.inputFilename);
		if (getenv("DSRC_HOST_DIRECT_IO") && S_ISREG(sb.st_mode))
		{	// the batches are read once, in order, by threads that copy into page-locked or 2 MiB-aligned buffers: on a fast NVMe
			// set the page cache is a second copy and an eviction cost.  Off by default; tmpfs and some overlays refuse O_DIRECT.
			fdDirect = open(args.inputFilename.c_str(), O_RDONLY | O_DIRECT);
			if (fdDirect < 0 && args.verboseLog) fprintf(stderr, "[dsrc-amd] DSRC_HOST_DIRECT_IO: this file system refuses O_DIRECT, reading through the page cache\n");
			else if (getenv("DSRC_HOST_TRACE")) fprintf(stderr, "[dsrc-amd] direct reads: %s\n", fdDirect >= 0 ? "on" : "refused");
		}
		if (!S_ISREG(sb.st_mode))
		{	// pipes and devices go through the stream reader
			close(fd); fd = -1;
			for (auto& t : warm) t.join();
			warm.clear();
			FILE* f = fopen(args.inputFilename.c_str(), "rb");
			if (!f) throw DsrcException("Cannot open file to read:" + args.inputFilename);
			const bool ok = ProcessStream(args, f);
			fclose(f);
			return ok;
		}
		const uint64 fileSize = (uint64)sb.st_size;
		settings = GetCompressionSettings(args);
		const uint64 bufSize = (uint64)args.fastqBufferSizeMB << 20;
		FileCutter cutter(fd, fileSize, bufSize);

		// first chunk: dataset analysis (FastqParser::Analyze)
		uint64 start0 = 0, size0 = 0;
		const bool findOffset = args.qualityOffset == fq::FastqDatasetType::AutoQualityOffset;
		if (!findOffset) type.qualityOffset = args.qualityOffset;
		{
			if (!cutter.Next(start0, size0)) throw DsrcException("Error analyzing FASTQ dataset");
			std::vector<uchar> first(size0);
			if (size0 && pread(fd, first.data(), size0, (off_t)start0) != (ssize_t)size0) throw DsrcException("read error");
			if (!AnalyzeFirstChunk(first.data(), first.size(), type, findOffset)) throw DsrcException("Error analyzing FASTQ dataset");
		}

		const uint32 batch = args.batchBlocks ? args.batchBlocks : (uint32)std::max<uint64>(1, (1536ull << 20) / bufSize);
		const uint64 nBatchesMax = (fileSize / bufSize + batch) / batch;
		// Several GPUs of one node: threadNum instances per device, all on the same work queue and the same
		// chain -- the chain lives in host memory, so the state of batch s reaches batch s+1 whichever device runs it, and
		// the archive is still the one `dsrc c -t1` writes.  No device-to-device traffic: blocks go to the ordered writer.
		const std::vector<int> devs = args.devices.empty() ? std::vector<int>(1, args.device) : args.devices;
		const uint32 instances = (uint32)std::max<uint64>(1, std::min<uint64>((uint64)std::min<uint32>(std::max(1u, args.threadNum), 8u) * devs.size(), nBatchesMax));
		if (dsrcgpu_chain_create(&chain) != DSRCGPU_OK) throw DsrcException("cannot create the batch chain");
		writer.Start(args.outputFilename);
		// typical archives are 0.2-0.35 of the text; what lies beyond the reservation is written like before
		writer.ReserveAhead(fileSize / 100 * (settings.lossy ? 28 : 36) + (64ull << 20));

		const bool trace = getenv("DSRC_HOST_TRACE") != nullptr;
		// ---- workers ------------------------------------------------------------------------------------------
		auto work = [&](uint32 idx)
		{
			dsrcgpu_handle* h = nullptr;
			// Two output buffers and a writer thread of the worker's own (round 5): the blocks of batch s go into the archive while the
			// instance already compresses its next batch (a worker that wrote its blocks itself spent 60-250 ms per 200-ms batch in the
			// write, profiles/r05_e2e_third.txt: all writers of one tmpfs file share the inode's lock).
			Pinned outs[2];
			struct WriteTask { std::unique_ptr<Job> job; Pinned* buf = nullptr; std::chrono::steady_clock::time_point t0, t1; };
			std::mutex wm; std::condition_variable wcv;
			std::deque<WriteTask> wq; bool wDone = false; bool busy[2] = {false, false};
			std::thread writerThread([&]()
			{
				try
				{
					for (;;)
					{
						WriteTask t;
						{
							std::unique_lock<std::mutex> g(wm);
							wcv.wait(g, [&] { return wDone || !wq.empty(); });
							if (wq.empty()) return;
							t = std::move(wq.front()); wq.pop_front();
						}
						Job* job = t.job.get();
						const uint32 n = (uint32)job->sizes.size();
						// the archive position of this batch is known once the batches before it have claimed theirs (sizes only:
						// nobody waits for anybody's bytes); the blocks lie back to back in the buffer and go out in one positioned write
						uint64 fileOff = 0, total = 0;
						bool failed = false;
						{
							std::unique_lock<std::mutex> g(pl.m);
							pl.cv.wait(g, [&] { return !pl.error.empty() || pl.claimTurn == job->seq; });
							failed = !pl.error.empty();
							if (!failed)
							{
								fileOff = writer.Claim(n, job->osz.data(), job->raw.data(), job->comp.data());
								++pl.claimTurn;
								pl.cv.notify_all();
							}
						}
						if (!failed)
						{
							for (uint32 i = 0; i < n; ++i) total += job->osz[i];
							writer.WriteAt(fileOff, t.buf->p + job->offs[0], total);
							std::lock_guard<std::mutex> g(pl.m);
							++pl.written; pl.cv.notify_all();
						}
						if (trace && !failed)
						{
							const auto t2 = std::chrono::steady_clock::now();
							auto ms = [&](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
							fprintf(stderr, "[dsrc-amd] batch %llu (%u chunks, %.2f GB): start %.0f ms, compress %.0f ms, written %.0f ms after that\n", (unsigned long long)job->seq, n, job->inBytes / 1e9, ms(tStart, t.t0), ms(t.t0, t.t1), ms(t.t1, t2));
						}
						{
							std::lock_guard<std::mutex> g(wm);
							busy[t.buf == &outs[1] ? 1 : 0] = false;
						}
						wcv.notify_all();
					}
				}
				catch (const std::exception& e) { pl.Fail(e.what()); std::lock_guard<std::mutex> g(wm); busy[0] = busy[1] = false; wcv.notify_all(); }
			});
			struct WriterJoin
			{
				std::thread& th; std::mutex& m; std::condition_variable& cv; bool& done;
				~WriterJoin() { { std::lock_guard<std::mutex> g(m); done = true; } cv.notify_all(); if (th.joinable()) th.join(); }
			} writerJoin{writerThread, wm, wcv, wDone};
			try
			{
				h = CreateInstance(args, settings, type, devs[idx % devs.size()]);
				if (trace) fprintf(stderr, "[dsrc-amd] instance %u ready at %.0f ms\n", idx, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tStart).count());
				for (uint32 turn = 0;; ++turn)
				{
					Job* job = nullptr;
					{	// batches start in order: the chain makes batch s+1 wait for batch s early in its course
						std::unique_lock<std::mutex> g(pl.m);
						pl.cv.wait(g, [&] { return !pl.error.empty() || pl.ready.count(pl.nextStart) || (pl.noMore && pl.nextStart >= totalBatches); });
						if (!pl.error.empty() || !pl.ready.count(pl.nextStart)) break;
						job = pl.ready[pl.nextStart]; pl.ready.erase(pl.nextStart); ++pl.nextStart;
						pl.cv.notify_all();
					}
					std::unique_ptr<Job> owner(job);
					Pinned& out = outs[turn & 1];
					{	// the buffer's previous batch is in the archive
						std::unique_lock<std::mutex> g(wm);
						wcv.wait(g, [&] { return !busy[turn & 1]; });
					}
					const auto t0 = std::chrono::steady_clock::now();
					const uint32 n = (uint32)job->sizes.size();
					const uint64 inBytes = job->inBytes;
					std::vector<const uint8_t*> ptrs(n);
					for (uint32 i = 0; i < n; ++i) ptrs[i] = job->in->p + job->at[i];
					uint64 cap = inBytes * 2 / 5 + (uint64)n * (1u << 16);     // typical ratio 0.2-0.33; grown below if the data needs it
					job->offs.resize(n); job->osz.resize(n); job->raw.resize(4 * n); job->comp.resize(4 * n);
					int rc;
					dsrcgpu_set_chain(h, chain, job->seq);     // once per batch: a retry below is the same turn of the chain
					for (;;)
					{
						out.Reserve(cap);
						rc = dsrcgpu_compress_batch(h, n, ptrs.data(), job->sizes.data(), out.p, out.cap, job->offs.data(), job->osz.data(), job->raw.data(), job->comp.data());
						if (rc != DSRCGPU_E_CAPACITY || cap >= inBytes + (uint64)n * (1u << 16)) break;
						cap = inBytes + (uint64)n * (1u << 16);        // incompressible input: room for the worst case, once
					}
					{
						std::lock_guard<std::mutex> g(pl.m);
						pl.freeIn.push_back(job->in); job->in = nullptr;
						pl.cv.notify_all();
					}
					if (rc != DSRCGPU_OK) throw DsrcException(dsrcgpu_last_error(h));
					{
						std::lock_guard<std::mutex> g(wm);
						busy[turn & 1] = true;
						wq.push_back(WriteTask{std::move(owner), &out, t0, std::chrono::steady_clock::now()});
					}
					wcv.notify_all();
				}
			}
			catch (const std::exception& e) { pl.Fail(e.what()); }
			if (h && !args.exitWhenDone) dsrcgpu_destroy(h);       // (a command-line process leaves its HBM to the exit: see DsrcDecompressorGPU::Process)
		};
		for (uint32 i = 0; i < instances; ++i) workers.emplace_back(work, i);

		// ---- readers: file -> page-locked memory, ahead of the scheduler instances -----------------------------------
		// A batch is read by SEVERAL threads (parts of 16 chunks): one thread copies out of the page cache at ~2 GB/s, so a
		// 1.6 GB batch read by one thread takes 0.9 s and (instances + 2) buffers in flight cap the pipeline at ~10 GB/s.
		const uint32 nReaders = std::min<uint32>(16, 4 * instances);
		const uint32 partChunks = 16;
		for (uint32 i = 0; i < instances + 2; ++i) { inBufs.emplace_back(new Pinned()); pl.freeIn.push_back(inBufs.back().get()); }
		auto readLoop = [&]()
		{
			try
			{
				for (;;)
				{
					Job* job = nullptr; uint32 lo = 0, hi = 0;
					{
						std::unique_lock<std::mutex> g(pl.m);
						for (;;)
						{
							if (!pl.error.empty()) return;
							// a part of a batch that is being read ...
							for (Job* j : pl.reading) if (j->nextPart < j->sizes.size()) { job = j; break; }
							if (job) { lo = job->nextPart; hi = std::min<uint32>((uint32)job->sizes.size(), lo + partChunks); job->nextPart = hi; break; }
							// ... or the next batch, once a buffer is free for it (its page-locked memory is sized here)
							if (!pl.todo.empty() && !pl.freeIn.empty())
							{
								Job* j = pl.todo.front(); pl.todo.pop_front();
								Pinned* buf = pl.freeIn.back(); pl.freeIn.pop_back();
								const uint32 n = (uint32)j->sizes.size();
								j->at.resize(n); j->inBytes = 0;
								// a chunk sits in its slot as far behind a 4 KiB boundary as it does in the file, so that a direct read
								// (aligned file offset, aligned address, whole sectors) can deliver it in place
								for (uint32 i = 0; i < n; ++i) { j->at[i] = j->inBytes + (j->fileOff[i] & 4095u); j->inBytes += (j->sizes[i] + 2 * 4096) & ~(uint64)4095; }
								j->in = buf; j->nextPart = n; j->chunksLeft = n;          // parts are handed out once the buffer exists
								pl.reading.push_back(j);
								g.unlock();
								buf->Reserve(j->inBytes);
								g.lock();
								j->nextPart = 0;
								pl.cv.notify_all();
								continue;
							}
							if (pl.noMore && pl.todo.empty() && pl.reading.empty()) return;
							pl.cv.wait(g);
						}
					}
					for (uint32 i = lo; i < hi; ++i)
					{
						uint64 got = 0;
						if (fdDirect >= 0)
						{	// the sectors that cover the chunk, past the page cache (DSRC_HOST_DIRECT_IO=1); whatever a direct read does
							// not deliver (end of file inside a sector, a file system that balks) comes through the ordinary descriptor
							const uint64 a0 = job->fileOff[i] & ~(uint64)4095, lead = job->fileOff[i] - a0;
							const uint64 a1 = (job->fileOff[i] + job->sizes[i] + 4095) & ~(uint64)4095;
							uchar* dst = job->in->p + job->at[i] - lead;
							uint64 have = 0;
							while (have < a1 - a0 && ((uintptr_t)(dst + have) & 4095u) == 0)
							{
								const ssize_t r = pread(fdDirect, dst + have, a1 - a0 - have, (off_t)(a0 + have));
								if (r <= 0) break;
								have += (uint64)r;
							}
							got = have > lead ? std::min<uint64>(have - lead, job->sizes[i]) : 0;
						}
						while (got < job->sizes[i])
						{
							const ssize_t r = pread(fd, job->in->p + job->at[i] + got, job->sizes[i] - got, (off_t)(job->fileOff[i] + got));
							if (r <= 0) throw DsrcException("read error");
							got += (uint64)r;
						}
					}
					std::lock_guard<std::mutex> g(pl.m);
					job->chunksLeft -= hi - lo;
					if (job->chunksLeft == 0)
					{
						pl.reading.erase(std::find(pl.reading.begin(), pl.reading.end(), job));
						if (trace && job->seq < 8) fprintf(stderr, "[dsrc-amd] batch %llu read at %.0f ms\n", (unsigned long long)job->seq, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tStart).count());
						pl.ready[job->seq] = job;
					}
					pl.cv.notify_all();
				}
			}
			catch (const std::exception& e) { pl.Fail(e.what()); }
		};
		for (uint32 i = 0; i < nReaders; ++i) readers.emplace_back(readLoop);

		// ---- cutter (this thread) -------------------------------------------------------------------------------
		uint64 seq = 0;
		bool more = true;
		uint64 start = start0, size = size0;
		bool havePending = true;
		while (more && !pl.Failed())
		{
			Job* job = new Job(); job->seq = seq;
			// (smaller first batches to fill the pipeline sooner were measured SLOWER: arenas and page-locked buffers sized by
			// a small first batch are re-allocated when the full-size batches arrive, 1.3-1.8 s per instance)
			while (job->sizes.size() < batch)
			{
				if (!havePending) { if (!cutter.Next(start, size)) { more = false; break; } }
				havePending = false;
				job->fileOff.push_back(start); job->sizes.push_back(size);
			}
			if (job->sizes.empty()) { delete job; break; }
			{
				std::unique_lock<std::mutex> g(pl.m);
				pl.cv.wait(g, [&] { return !pl.error.empty() || pl.todo.size() < 2 * instances; });
				pl.todo.push_back(job); ++seq;
				pl.cv.notify_all();
			}
		}
		{
			std::lock_guard<std::mutex> g(pl.m);
			pl.noMore = true; totalBatches = seq;
			pl.cv.notify_all();
		}
		auto stamp = [&](const char* what) { if (trace) fprintf(stderr, "[dsrc-amd] %s at %.0f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tStart).count()); };
		stamp("cutter done");
		{	// every block is in the file once all batches have been written; the instances may still be tearing down
			std::unique_lock<std::mutex> g(pl.m);
			pl.cv.wait(g, [&] { return !pl.error.empty() || pl.written == totalBatches; });
		}
		if (!pl.error.empty()) throw DsrcException(pl.error);
		writer.Finish(type, settings);
		LogSizes(writer);
		stamp("archive closed");
		if (args.exitWhenDone)
		{	// a command-line process has nothing left to do: the instances have handed their HBM back in their own threads, the batch
			// buffers go here, from several threads (pageable ones only: a page-locked buffer belongs to the runtime)
			// (dropping the batch buffers from eight threads first -- madvise(MADV_DONTNEED) -- was measured: 0.45 s for that and the
			// exit no shorter, profiles/r05_e2e_5.txt)
			if (args.verboseLog) fputs(GetLog().c_str(), stderr);
			fflush(nullptr);
			_exit(0);
		}
		for (auto& t : workers) t.join();
		for (auto& t : readers) t.join();
		workers.clear(); readers.clear();
		inBufs.clear();
		stamp("instances and buffers released");
	}
	catch (const DsrcException& e) { AddError(e.what()); pl.Fail(e.what()); }
	catch (const std::exception& e) { AddError(e.what()); pl.Fail(e.what()); }
	{
		std::lock_guard<std::mutex> g(pl.m);
		pl.noMore = true; pl.cv.notify_all();
	}
	for (auto& t : warm) if (t.joinable()) t.join();
	for (auto& t : workers) if (t.joinable()) t.join();
	for (auto& t : readers) if (t.joinable()) t.join();
	for (Job* j : pl.todo) delete j;
	for (Job* j : pl.reading) delete j;
	for (auto& kv : pl.ready) delete kv.second;
	if (chain) dsrcgpu_chain_destroy(chain);
	if (fd >= 0) close(fd);
	if (fdDirect >= 0) close(fdDirect);
	return !IsError();
}


// ---- archive reader (DsrcFileReader, src/DsrcFile.cpp:172-318) -------------------------------------------------------------
uint64 GetBE(const uchar* p, int bytes) { uint64 v = 0; for (int i = 0; i < bytes; ++i) v = (v << 8) | p[i]; return v; }

ArchiveReader::~ArchiveReader() { Close(); }
void ArchiveReader::Close() { if (fd >= 0) { close(fd); fd = -1; } }

void ArchiveReader::Open(const std::string& path)
{
	fd = open(path.c_str(), O_RDONLY);
	struct stat sb;
	if (fd < 0 || fstat(fd, &sb) != 0) { Close(); throw DsrcException("Cannot open file to read:" + path); }
	const uint64 fileSize = (uint64)sb.st_size;
	if (fileSize == 0) { Close(); throw DsrcException("Empty file."); }
	uchar head[40];
	if (fileSize < 40 || pread(fd, head, 40, 0) != 40 || !(head[1] == 2 && head[2] == 0))     // VersionMajor.VersionMinor = 2.0 (src/DsrcFile.h:31-33)
	{ Close(); throw DsrcException("Invalid archive or old unsupported version"); }
	const uint64 footerSize = GetBE(head + 4, 4), footerOffset = GetBE(head + 8, 8), blockCount = GetBE(head + 24, 8);
	if (blockCount == 0 || footerOffset + footerSize > fileSize || footerOffset < 40 || footerSize < 1 + 4 * blockCount + 13 || blockCount > fileSize / 4)
	{ Close(); throw DsrcException("Corrupted DSRC archive header"); }
	std::vector<uchar> foot(footerSize);
	if (pread(fd, foot.data(), footerSize, (off_t)footerOffset) != (ssize_t)footerSize || foot[0] != 0xCC)
	{ Close(); throw DsrcException("Corrupted DSRC archive footer"); }
	blockSizes.resize(blockCount);
	memcpy(blockSizes.data(), foot.data() + 1, blockCount * 4);          // host-endian uint32 array, as the reference reads it
	const uchar* q = foot.data() + 1 + blockCount * 4;
	type.colorSpace = (q[0] & 2) != 0; type.plusRepetition = (q[0] & 1) != 0; type.qualityOffset = q[1];
	settings.lossy = (q[2] & 1) != 0; settings.calculateCrc32 = (q[2] & 2) != 0;
	settings.dnaOrder = q[3]; settings.qualityOrder = q[4]; settings.tagPreserveFlags = GetBE(q + 5, 8);
	blockOffs.resize(blockCount);
	uint64 at = 40;
	for (uint64 i = 0; i < blockCount; ++i) { blockOffs[i] = at; at += blockSizes[i]; }
	if (at > footerOffset) { Close(); throw DsrcException("Corrupted DSRC archive footer"); }
}

void ArchiveReader::ReadBlock(uint64 i, uchar* dst) const
{
	uint64 got = 0;
	while (got < blockSizes[i])
	{
		const ssize_t r = pread(fd, dst + got, blockSizes[i] - got, (off_t)(blockOffs[i] + got));
		if (r <= 0) throw DsrcException("read error");
		got += (uint64)r;
	}
}

// Text bytes to reserve for consecutive blocks.  A block written from a FASTQ file declares its own chunk size (+1 for the
// last newline, src/BlockCompressor.cpp:279-281); one written by the record-level API declares a running total over the
// archive (BlockCompressor::Reset does not clear it, src/BlockCompressorExt.cpp:126), i.e. its own size is the difference
// to the block before.  The difference is taken for the size only where it is plausible -- smaller than the word itself and
// not smaller than the block's compressed size (consecutive chunks of a file archive differ by a few KB, a block of text
// is larger than its compressed form).  `exact` = false asks for the always-sufficient figure (used when the first try
// did not fit).
void TextCaps(const std::vector<uint32>& words, const std::vector<uint64_t>& blockSizes, uint32 wordBefore, bool haveBefore, bool exact, std::vector<uint64_t>& caps)
{
	caps.resize(words.size());
	uint32 prev = wordBefore; bool have = haveBefore;
	for (size_t i = 0; i < words.size(); ++i)
	{
		const uint64 own = (uint64)words[i] + 1;
		const uint64 diff = (uint64)(uint32)(words[i] - prev) + 1;
		// second try: whichever is larger (a running total that has wrapped past 4 GiB makes `own` the small one)
		caps[i] = exact ? ((have && diff < own && diff >= blockSizes[i]) ? diff : own) : std::max(own, have ? diff : own);
		prev = words[i]; have = true;
	}
}

dsrcgpu_handle* CreateDecodeInstance(int device, const CompressionSettings& settings, const fq::FastqDatasetType& type)
{
	InputParameters a; a.device = device;
	return DsrcCompressorGPU::CreateInstance(a, settings, type);
}

// Archive -> FASTQ.  `instances` workers each own one GPU scheduler instance; a worker takes the next batch of blocks,
// reads it from the archive into page-locked memory, decompresses it and writes the text when the batches before it have
// been written (one worker reads while another decodes while a third writes).
bool DsrcDecompressorGPU::Process(const InputParameters& args)
{
	FILE* out = nullptr;
	ArchiveReader rd;
	std::vector<std::thread> workers;
	std::mutex m; std::condition_variable cv;
	std::string error;
	const auto tProc = std::chrono::steady_clock::now();
	std::vector<std::thread> warm;                   // HIP start-up runs beside the opening of the archive and the sizing of the output
	struct WarmGuard { std::vector<std::thread>& th; ~WarmGuard() { for (auto& t : th) if (t.joinable()) t.join(); } } warmGuard{warm};
	try
	{
		{
			const std::vector<int> d0 = args.devices.empty() ? std::vector<int>(1, args.device) : args.devices;
			for (int dev : d0) warm.emplace_back([dev]() { (void)dsrcgpu_prepare(dev); });
		}
		rd.Open(args.inputFilename);
		if (args.useFastqStdIo) out = stdout;
		else { out = fopen(args.outputFilename.c_str(), "w+b"); if (!out) throw DsrcException("Cannot open file to write:" + args.outputFilename); }      // readable too: the output is mapped
		// Only a regular file can be sized, mapped and written at positions; anything else that was named as the output (a pipe,
		// /dev/stdout, a process substitution) gets its bytes in batch order through fwrite, like stdout and like the reference
		bool regular = false;
		if (out != stdout) { struct stat sb; regular = fstat(fileno(out), &sb) == 0 && S_ISREG(sb.st_mode); }

		const uint64 nBlocks = rd.BlockCount();
		// A decoding pass is a chain per block: it takes about as long for 1000 blocks as for 10, so
		// passes are LARGE -- up to 4800 blocks, 14 GiB of archive -- and few handles run at a time: with an order model every
		// block in flight holds a model table of up to 64 MiB, and the handles of a device share the HBM for them.
		const std::vector<int> devs = args.devices.empty() ? std::vector<int>(1, args.device) : args.devices;
		const bool tables = rd.Settings().dnaOrder > 0 || rd.Settings().qualityOrder > 0;
		// (round 5, with the lean quality loop: the 37.7 GB set in two passes of 2250 blocks 8.2-8.3 s, in three of 1500 8.2-9.4 s, in one 8.0-8.2 s)
		const uint32 perDev = std::min<uint32>(std::max(1u, args.threadNum), tables ? 2u : 4u);
		const uint32 wanted = perDev * (uint32)devs.size();
		std::vector<std::pair<uint64, uint64> > batches;
		{
			// as few rounds of `wanted` concurrent passes as passes of <= 4800 blocks (14 GiB of archive) allow, all of one size ...
			// ... and as the device's memory allows.  What a pass keeps in HBM per block: the block itself, its text (the chunk it was cut
			// from: 4 - 7 x the block on FASTQ data, taken as 8 x), the decoder's arena (3 x the block + 1 MiB) and, with an order model,
			// its share of the table region (16 MiB).  `perDev` passes run side by side on a device and together stay below 85 % of what
			// is free now; the table region is capped by the library as well (dsrcgpu_set_table_budget below).
			uint64 memBlocks = 4800;
			{
				uint64_t freeB = 0, totalB = 0;
				uint64 archive = 0; for (uint64 i = 0; i < nBlocks; ++i) archive += rd.BlockSizes()[i];
				const uint64 avg = nBlocks ? archive / nBlocks + 1 : 1;
				const uint64 perBlock = avg * (1 + 8 + 3) + (1ull << 20) + (tables ? (16ull << 20) : 0);
				if (dsrcgpu_device_memory(devs[0], &freeB, &totalB) == DSRCGPU_OK && freeB)
					memBlocks = std::max<uint64>(16, std::min<uint64>(4800, freeB / 100 * 85 / perDev / perBlock));
			}
			const uint64 rounds = std::max<uint64>(1, (nBlocks + (uint64)wanted * memBlocks - 1) / ((uint64)wanted * memBlocks));
			const uint64 maxBlocks = args.batchBlocks ? args.batchBlocks : std::max<uint64>(1, (nBlocks + wanted * rounds - 1) / (wanted * rounds));
			const uint64 budget = 14336ull << 20;
			uint64 lo = 0, bytes = 0;
			for (uint64 i = 0; i < nBlocks; ++i)
			{
				bytes += rd.BlockSizes()[i];
				if (i + 1 - lo >= maxBlocks || bytes >= budget) { batches.emplace_back(lo, i + 1); lo = i + 1; bytes = 0; }
			}
			if (lo < nBlocks) batches.emplace_back(lo, nBlocks);
		}
		const uint32 instances = (uint32)std::max<uint64>(1, std::min<uint64>(wanted, batches.size()));
		uint64 tableShare = 0;
		if (tables)
		{
			uint64_t freeB = 0, totalB = 0;
			if (dsrcgpu_device_memory(devs[0], &freeB, &totalB) == DSRCGPU_OK) tableShare = freeB / 100 * 60 / std::max<uint32>(1, (instances + (uint32)devs.size() - 1) / (uint32)devs.size());
		}
		uint64 next = 0, writeTurn = 0, outPos = 0, mappedDone = 0;

		// ---- a regular file as output: decode straight into its mapping -----------------------------------------------------------
		// A block written from a FASTQ file declares its chunk size, so where every block's text goes is known before anything is
		// decoded: the file is sized and mapped, helper threads fault its pages in (in batch order) while the GPU decodes, and the
		// device-to-host copy of a batch lands in the file -- no text buffer, no second copy, no write.  Archives whose chunk-size
		// words are running totals (record-level API) or whose texts come out shorter than declared take the buffered path below.
		uchar* map = nullptr; uint64 mapBytes = 0;
		std::vector<uint64> mapAt(batches.size(), 0);
		std::vector<std::vector<uint64_t> > mapCaps(batches.size());
		std::vector<std::thread> faulters, unmappers;
		std::atomic<bool> mapBroken(false), faultStop(false);
		std::atomic<uint64> reservedUpTo(0);             // bytes of the output file whose pages exist (fallocate, in file order)
		uint32 instancesReady = 0;                       // (m) workers whose scheduler instance exists: the helper threads start behind them
		struct MapGuard          // whatever way this scope is left: helper threads joined, mapping gone
		{
			std::vector<std::thread>& th; std::vector<std::thread>& th2; uchar*& p; uint64& n;
			std::atomic<bool>& stop; std::condition_variable& cv;
			~MapGuard()
			{
				stop.store(true); cv.notify_all();
				for (auto& t : th) if (t.joinable()) t.join();
				for (auto& t : th2) if (t.joinable()) t.join();
				if (p) munmap(p, n);
				p = nullptr;
			}
		} mapGuard{faulters, unmappers, map, mapBytes, faultStop, cv};
		if (regular && !getenv("DSRC_HOST_NO_MMAP") && nBlocks)
		{
			std::vector<uint32> words(nBlocks); std::vector<uint64_t> bsz(nBlocks);
			bool ok = true;
			for (uint64 i = 0; i < nBlocks && ok; ++i)
			{
				uchar hb[16]; bsz[i] = rd.BlockSizes()[i];
				ok = bsz[i] >= 16 && pread(rd.Fd(), hb, 16, (off_t)rd.BlockOffset(i)) == 16;
				if (ok) words[i] = (uint32)GetBE(hb + 12, 4);
			}
			// chunks cut from a file are all of one size within the 8 KiB the cutter looks ahead plus a record (src/FastqStream.cpp:18-72);
			// running totals (record-level archives) grow by a chunk per block: those are not sizes
			if (ok && nBlocks >= 2)
			{
				uint32 lo = words[0], hi = words[0];
				for (uint64 i = 1; i + 1 < nBlocks; ++i) { lo = std::min(lo, words[i]); hi = std::max(hi, words[i]); }
				if (hi - lo > (64u << 10) || words[nBlocks - 1] > hi + (64u << 10)) ok = false;
			}
			if (ok)
			{
				uint64 total = 0;
				for (size_t k = 0; k < batches.size(); ++k)
				{
					mapAt[k] = total;
					for (uint64 i = batches[k].first; i < batches[k].second; ++i) { mapCaps[k].push_back((uint64)words[i] + 1); total += (uint64)words[i] + 1; }
				}
				// The space is reserved, not just declared: a full disk shows as a failed fallocate (and the buffered path reports it)
				// instead of as a SIGBUS in the middle of a copy into the mapping; a file system without fallocate takes the buffered
				// path as well.  Round 5: the reservation no longer stands in front of everything -- one helper thread reserves the
				// file 256 MiB at a time in file order (tmpfs: 18 GB/s, the whole 37.7 GB set in 2.1 s, while the HIP runtime comes up and
				// the first passes run) and a batch waits for ITS range only; and the threads that map the pages into this process start
				// once every scheduler instance exists (six of them hammering the address space from the first millisecond had held the
				// runtime's own mmaps back: an instance was ready after 1.46 s instead of 0.16 s, profiles/r05_e2e_first.txt).
				const uint64 first = std::min<uint64>(total, 256ull << 20);
				if (total && fallocate(fileno(out), 0, 0, (off_t)first) == 0)
				{
					void* q = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fileno(out), 0);
					if (q != MAP_FAILED)
					{
						map = (uchar*)q; mapBytes = total;
						reservedUpTo.store(first);
						faulters.emplace_back([&, total, first]()
						{
							const uint64 piece = 256ull << 20;
							for (uint64 o = first; o < total && !faultStop.load(); o += piece)
							{
								const uint64 len = std::min<uint64>(piece, total - o);
								if (fallocate(fileno(out), 0, (off_t)o, (off_t)len) != 0)
								{	// disk full (or the file system changed its mind): nobody may touch the mapping beyond this point
									std::lock_guard<std::mutex> g(m);
									mapBroken = true; cv.notify_all();
									return;
								}
								{ std::lock_guard<std::mutex> g(m); reservedUpTo.store(o + len); }
								cv.notify_all();
							}
						});
						auto fault = [&, total](uint32 t, uint32 nt)
						{	// 32 MiB pieces (a piece holds the address space's lock against the runtime's mmaps for a few ms), in file order,
							// round-robin over the threads, behind the reservation
							{	// behind the runtime's start-up and behind the WHOLE reservation: mapping pages in while fallocate hands them out
								// slowed the reservation from 18 to 4.5 GB/s (the third batch waited 8.4 s for its range, profiles/r05_e2e_second.txt)
								std::unique_lock<std::mutex> g(m);
								cv.wait(g, [&] { return faultStop.load() || mapBroken.load() || !error.empty() || (instancesReady >= instances && reservedUpTo.load() >= total); });
							}
							const uint64 piece = 32ull << 20;
							for (uint64 o = (uint64)t * piece; o < total; o += (uint64)nt * piece)
							{
								const uint64 len = std::min<uint64>(piece, total - o);
								{
									std::unique_lock<std::mutex> g(m);
									cv.wait(g, [&] { return faultStop.load() || mapBroken.load() || reservedUpTo.load() >= o + len; });
									if (faultStop.load() || mapBroken.load()) return;
								}
#ifdef MADV_POPULATE_WRITE
								if (madvise(map + o, len, MADV_POPULATE_WRITE) == 0) continue;
								if (errno != EINVAL) continue;       // (ENOMEM: a batch that is through has given its range back already)
#endif
								if (args.exitWhenDone) continue;      // ... which is why nothing is touched by hand in that mode
								for (uint64 x = 0; x < len; x += 4096) (void)((volatile const uchar*)map)[o + x];      // a read: the decoded bytes may be there already
							}
						};
						for (uint32 t = 0; t < 6; ++t) faulters.emplace_back(fault, t, 6u);
					}
					else if (ftruncate(fileno(out), 0) != 0) throw DsrcException("Error writing FASTQ output");
				}
				else if (total && ftruncate(fileno(out), 0) != 0) throw DsrcException("Error writing FASTQ output");
			}
		}

		const bool trace = getenv("DSRC_HOST_TRACE") != nullptr;
		const auto t0 = tProc;
		if (trace) fprintf(stderr, "[dsrc-amd d] archive open, output sized and mapped %8.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
		auto mark = [&](uint32 idx, uint64 k, const char* what)
		{
			if (trace) fprintf(stderr, "[dsrc-amd d] worker %u batch %llu %-10s %8.1f ms\n", idx, (unsigned long long)k, what,
							   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
		};
		// one handle per worker, kept across the two rounds of an archive whose mapped attempt is abandoned: a second set of
		// instances would claim the arenas and the model-table share a second time
		std::vector<dsrcgpu_handle*> handles(instances, nullptr);
		auto work = [&](uint32 idx)
		{
			dsrcgpu_handle*& h = handles[idx];
			try
			{
				if (!h)
				{
					h = CreateDecodeInstance(devs[idx % devs.size()], rd.Settings(), rd.Type());
					mark(idx, 0, "instance");
				}
				{ std::lock_guard<std::mutex> g(m); ++instancesReady; cv.notify_all(); }
				if (tableShare) dsrcgpu_set_table_budget(h, tableShare);
				Pinned in, text;
				// device buffers of the mapped road (blocks in, text out), kept from pass to pass
				struct DevBuf
				{
					dsrcgpu_handle*& h; void* p = nullptr; uint64 cap = 0;
					bool Reserve(uint64 n) { if (n <= cap) return true; Release(); if (dsrcgpu_dev_alloc(h, n + n / 32, &p) != DSRCGPU_OK) { p = nullptr; return false; } cap = n + n / 32; return true; }
					void Release() { if (p) dsrcgpu_dev_free(h, p); p = nullptr; cap = 0; }
					~DevBuf() { Release(); }
				} dBlocks{h}, dText{h};
				for (;;)
				{
					uint64 k;
					{
						std::lock_guard<std::mutex> g(m);
						if (!error.empty() || next >= batches.size()) break;
						k = next++;
					}
					const uint64 lo = batches[k].first, hi = batches[k].second; const uint32 n = (uint32)(hi - lo);
					std::vector<uint64_t> sizes(n), at(n), offs(n), tsz(n), caps;
					uint64 inBytes = 0;
					for (uint32 i = 0; i < n; ++i) { sizes[i] = rd.BlockSizes()[lo + i]; at[i] = inBytes; inBytes += (sizes[i] + 64) & ~(uint64)63; }
					mark(idx, k, "start");
					in.Reserve(inBytes);
					std::thread reserveHbm;
					if (map && !mapBroken)
					{	// what the pass will take of the device is asked for now, beside the reading of the archive: HBM that the previous
						// process has just released is wiped before it is handed out, and the pass would wait for that in its middle
						uint64 cap = 0; for (uint64 c : mapCaps[k]) cap += c;
						const uint64 tab = rd.Settings().qualityOrder || rd.Settings().dnaOrder ? (uint64)n * (16ull << 20) : 0;
						reserveHbm = std::thread([&, cap, tab]()
						{
							(void)dText.Reserve(cap + 64);
							(void)dsrcgpu_reserve_memory(h, inBytes * 3 + (uint64)n * (1u << 20) + (32u << 20), tab);
						});
					}
					struct JoinGuard { std::thread& t; ~JoinGuard() { if (t.joinable()) t.join(); } } reserveJoin{reserveHbm};
					std::vector<const uint8_t*> ptrs(n); std::vector<uint32> words(n);
					// the mapped road sends the blocks up in eight parts, each as soon as its blocks have been read
					const bool mappedRoad = map && !mapBroken;
					bool uploaded = false;
					{	// the blocks of the batch, read by up to 32 threads (positioned reads; 12 GB by 2 x 16 threads took 0.85 s of an 8.2 s run)
						constexpr uint32 PARTS = 8;
						std::atomic<uint32> nextBlock(0); std::atomic<bool> bad(false);
						std::atomic<uint32> partDone[PARTS];
						for (auto& x : partDone) x.store(0);
						const uint32 per = (n + PARTS - 1) / PARTS;
						std::mutex um; std::condition_variable ucv;
						auto get = [&]()
						{
							try
							{
								for (uint32 i; (i = nextBlock++) < n;)
								{
									rd.ReadBlock(lo + i, in.p + at[i]);
									const uint32 pt = i / per, inPart = std::min(n, (pt + 1) * per) - pt * per;
									if (++partDone[pt] == inPart) { std::lock_guard<std::mutex> g(um); ucv.notify_all(); }
								}
							}
							catch (...) { bad = true; std::lock_guard<std::mutex> g(um); ucv.notify_all(); }
						};
						std::vector<std::thread> rs;
						for (uint32 t = 0; t < std::min<uint32>(32, n); ++t) rs.emplace_back(get);
						if (mappedRoad && dBlocks.Reserve(inBytes + 64))
						{
							for (uint32 pt = 0; pt * per < n && !bad; ++pt)
							{
								const uint32 b0 = pt * per, b1 = std::min(n, b0 + per);
								{
									std::unique_lock<std::mutex> g(um);
									ucv.wait(g, [&] { return bad.load() || partDone[pt].load() == b1 - b0; });
								}
								if (bad) break;
								const uint64 o0 = at[b0], o1 = b1 < n ? at[b1] : inBytes;
								if (dsrcgpu_dev_upload(h, (uchar*)dBlocks.p + o0, in.p + o0, o1 - o0) != DSRCGPU_OK) { bad = true; break; }
							}
							uploaded = !bad;
						}
						for (auto& t : rs) t.join();
						if (bad) throw DsrcException("Error reading the DSRC archive");
					}
					for (uint32 i = 0; i < n; ++i)
					{
						ptrs[i] = in.p + at[i];
						if (sizes[i] < 16) throw DsrcException("Corrupted DSRC archive: block too small");
						words[i] = (uint32)GetBE(ptrs[i] + 12, 4);
					}
					// the block before this batch tells whether chunk sizes are running totals (record-level archives)
					uint32 before = 0; bool haveBefore = false;
					if (lo > 0)
					{
						uchar hb[16];
						if (rd.BlockSizes()[lo - 1] >= 16 && pread(rd.Fd(), hb, 16, (off_t)rd.BlockOffset(lo - 1)) == 16) { before = (uint32)GetBE(hb + 12, 4); haveBefore = true; }
					}
					if (reserveHbm.joinable()) reserveHbm.join();
					mark(idx, k, "read");
					int rc = DSRCGPU_OK;
					if (mappedRoad && !mapBroken)
					{
						// Round 5: the pass is device-resident -- blocks up, decode, text down as three calls of the C ABI -- so that the
						// only thing that needs this batch's range of the output file, the copy down, is the only thing that waits for it
						// (the reservation of the whole file takes 2 s; the third batch used to start 1.3 s late for it)
						uint64 cap = 0; for (uint64 c : mapCaps[k]) cap += c;
						if (!dBlocks.Reserve(inBytes + 64) || !dText.Reserve(cap + 64)) throw DsrcException(dsrcgpu_last_error(h));
						if (!uploaded && dsrcgpu_dev_upload(h, dBlocks.p, in.p, inBytes) != DSRCGPU_OK) throw DsrcException(dsrcgpu_last_error(h));
						mark(idx, k, "uploaded");
						rc = dsrcgpu_decompress_batch_device(h, n, dBlocks.p, at.data(), sizes.data(), mapCaps[k].data(), dText.p, cap, offs.data(), tsz.data(), nullptr);
						bool exact = rc == DSRCGPU_OK;
						for (uint32 i = 0; i < n && exact; ++i) exact = tsz[i] == mapCaps[k][i];
						if (exact)
						{
							mark(idx, k, "decoded");
							{	// the pages of this batch's range exist (see the reservation above)
								std::unique_lock<std::mutex> g(m);
								cv.wait(g, [&] { return mapBroken.load() || !error.empty() || reservedUpTo.load() >= mapAt[k] + cap; });
								if (!error.empty()) break;
							}
							if (mapBroken) throw DsrcException("__remap__");
							if (dsrcgpu_dev_download(h, map + mapAt[k], dText.p, cap) != DSRCGPU_OK) throw DsrcException(dsrcgpu_last_error(h));
							mark(idx, k, "copied");
							// (Giving a finished batch's range of the mapping back at once, on a thread of its own, was measured: munmap of 18.8 GB of
							// dirty shared pages takes 1.4 s there against 0.55 s of the process's exit, profiles/r05_e2e_6_t2.txt.)
							std::lock_guard<std::mutex> g(m);
							++mappedDone; cv.notify_all();
							continue;
						}
						if (rc != DSRCGPU_OK && rc != DSRCGPU_E_CAPACITY) throw DsrcException(dsrcgpu_last_error(h));
						// the declared sizes are not where the texts end: this archive goes the buffered way, from the start
						std::lock_guard<std::mutex> g(m);
						mapBroken = true; cv.notify_all();
						throw DsrcException("__remap__");
					}
					for (int attempt = 0; attempt < 2; ++attempt)
					{
						TextCaps(words, sizes, before, haveBefore, attempt == 0, caps);
						uint64 cap = 0; for (uint64 c : caps) cap += c;
						text.Reserve(cap + 64);
						mark(idx, k, "reserved");
						rc = dsrcgpu_decompress_batch(h, n, ptrs.data(), sizes.data(), caps.data(), text.p, text.cap, offs.data(), tsz.data(), nullptr);
						if (rc != DSRCGPU_E_CAPACITY) break;
					}
					if (rc != DSRCGPU_OK) throw DsrcException(dsrcgpu_last_error(h));
					mark(idx, k, "decoded");
					uint64 total = 0; for (uint32 i = 0; i < n; ++i) total += tsz[i];
					uint64 fileAt = 0;
					{
						std::unique_lock<std::mutex> g(m);
						cv.wait(g, [&] { return !error.empty() || writeTurn == k; });
						if (!error.empty()) break;
						// a file: the batch claims its range (that needs the sizes of the batches before it, not their bytes) and
						// passes the turn on at once; a pipe: the bytes go out in turn
						if (regular) { fileAt = outPos; outPos += total; ++writeTurn; cv.notify_all(); }
					}
					// texts are laid out back to back; they are contiguous whenever every block fills its reservation
					std::vector<std::pair<uint64, uint64> > runs;           // (offset in text.p, length)
					for (uint32 i = 0; i < n;)
					{
						uint32 j = i; uint64 len = tsz[i];
						while (j + 1 < n && offs[j] + tsz[j] == offs[j + 1]) { ++j; len += tsz[j]; }
						if (len) runs.emplace_back(offs[i], len);
						i = j + 1;
					}
					if (!regular)
					{
						for (const auto& r : runs)
							if (fwrite(text.p + r.first, 1, r.second, out) != r.second) throw DsrcException("Error writing FASTQ output (disk full?)");
						std::lock_guard<std::mutex> g(m);
						++writeTurn; cv.notify_all();
					}
					else
					{	// positioned writes, 32 MiB pieces, by up to 8 threads (one thread copies into the page cache at ~2 GB/s)
						struct Piece { const uchar* p; uint64 n; uint64 at; };
						std::vector<Piece> pieces;
						uint64 at = fileAt;
						for (const auto& r : runs)
						{
							for (uint64 o = 0; o < r.second; o += 32ull << 20) pieces.push_back({text.p + r.first + o, std::min<uint64>(32ull << 20, r.second - o), at + o});
							at += r.second;
						}
						std::atomic<size_t> nextPiece(0); std::atomic<bool> bad(false);
						auto put = [&]()
						{
							for (size_t q; (q = nextPiece++) < pieces.size();)
							{
								const Piece& pc = pieces[q];
								for (uint64 done = 0; done < pc.n;)
								{
									const ssize_t w = pwrite(fileno(out), pc.p + done, pc.n - done, (off_t)(pc.at + done));
									if (w <= 0) { bad = true; return; }
									done += (uint64)w;
								}
							}
						};
						std::vector<std::thread> ws;
						for (size_t t = 1; t < std::min<size_t>(8, pieces.size()); ++t) ws.emplace_back(put);
						put();
						for (auto& t : ws) t.join();
						if (bad) throw DsrcException("Error writing FASTQ output (disk full?)");
						mark(idx, k, "written");
					}
				}
			}
			catch (const std::exception& e) { std::lock_guard<std::mutex> g(m); if (error.empty()) error = e.what(); cv.notify_all(); }
			// (Handing the worker's HBM back here, before the process leaves, was measured: 0.3-0.5 s for the calls and the exit -- 1.4 s,
			// the output mapping's page tables -- no shorter: profiles/r05_e2e_third.txt against r05_e2e_4_t3.txt.)
		};
		// a command-line process that is about to leave keeps its handles (the memory behind them has gone back above)
		struct HandleGuard
		{
			std::vector<dsrcgpu_handle*>& hs; std::vector<std::thread>& th; bool keep;
			~HandleGuard() { for (auto& t : th) if (t.joinable()) t.join(); if (!keep) for (auto*& h : hs) if (h) { dsrcgpu_destroy(h); h = nullptr; } }
		} handleGuard{handles, workers, args.exitWhenDone};
		for (uint32 i = 0; i < instances; ++i) workers.emplace_back(work, i);
		for (auto& t : workers) t.join();
		workers.clear();
		mark(0, batches.size(), "all decoded");
		faultStop.store(true); cv.notify_all();          // whatever has not been mapped by now is of no use to anybody
		for (auto& t : faulters) t.join();
		faulters.clear();
		mark(0, batches.size(), "helpers joined");
		if (args.exitWhenDone && error.empty() && !mapBroken && regular)
		{	// the text is in the page cache (through the mapping or pwrite); unmapping, closing and the HIP teardown are the
			// kernel's job at exit, where nobody waits for them one after the other
			if (args.verboseLog) fputs(GetLog().c_str(), stderr);
			fflush(nullptr);
			_exit(0);
		}
		if (map)
		{
			munmap(map, mapBytes); map = nullptr;
			if (mapBroken)
			{	// once more without the mapping (record-level archives, texts shorter than declared)
				if (ftruncate(fileno(out), 0) != 0) throw DsrcException("Error writing FASTQ output");
				error.clear(); next = 0; writeTurn = 0; outPos = 0;
				for (uint32 i = 0; i < instances; ++i) workers.emplace_back(work, i);
				for (auto& t : workers) t.join();
				workers.clear();
			}
		}
		if (!error.empty()) throw DsrcException(error);
		if (out != stdout) { FILE* f = out; out = nullptr; if (fclose(f) != 0) throw DsrcException("Error writing FASTQ output (disk full?)"); }
		else fflush(stdout);
	}
	catch (const DsrcException& e) { AddError(e.what()); }
	catch (const std::exception& e) { AddError(e.what()); }
	for (auto& t : workers) if (t.joinable()) t.join();
	if (out && out != stdout) { fclose(out); if (IsError()) unlink(args.outputFilename.c_str()); }
	return !IsError();
}

} // namespace comp

namespace wrap
{
// the reference's setters, their ranges and their texts (src/Configurable.cpp:56-144): the command line has checks of its own
// (src/main.cpp:276-297: quality offset 33..64 or auto, 1..64 threads) which live in main.cpp here as there
void Configurable::SetFastqBufferSizeMB(uint64 s) { if (s == 0 || s > 1024) throw DsrcException("Invalid argument: invalid FASTQ buffer size [1-1024]"); params.fastqBufferSizeMB = (uint32)s; }
void Configurable::SetDnaCompressionLevel(uint32 l) { if (l > 3) throw DsrcException("Invalid argument: invalid DNA compression level [0-3]"); params.dnaCompressionLevel = l; }
void Configurable::SetQualityCompressionLevel(uint32 l) { if (l > 2) throw DsrcException("Invalid argument: invalid Quality compression level [0-2]"); params.qualityCompressionLevel = l; }
void Configurable::SetQualityOffset(uint32 o) { if (o != 33 && o != 64) throw DsrcException("Invalid argument: only valid Quality offset are 33 and 64"); params.qualityOffset = o; }
void Configurable::SetThreadsNumber(uint32 t) { if (t == 0) throw DsrcException("Invalid argument: thread number must be greater than 0"); params.threadNum = t; }

void DsrcModule::Compress(const std::string& in, const std::string& out)
{
	comp::DsrcCompressorGPU op;
	comp::InputParameters p = params;
	p.inputFilename = in; p.outputFilename = out;
	if (!op.Process(p)) throw DsrcException(op.GetError());
}

// ---- FastqFile --------------------------------------------------------------------------------------------------
FastqFile::~FastqFile() { if (file) fclose(file); }

void FastqFile::Open(const std::string& filename_)
{
	if (file) throw DsrcException("Invalid state");
	file = fopen(filename_.c_str(), "rb");
	if (!file) throw DsrcException(("Cannot open file to read:" + filename_).c_str());
	writing = false;
}

void FastqFile::Create(const std::string& filename_)
{
	if (file) throw DsrcException("Invalid state");
	file = fopen(filename_.c_str(), "wb");
	if (!file) throw DsrcException(("Cannot open file to write:" + filename_).c_str());
	writing = true;
}

void FastqFile::Close()
{
	if (!file) throw DsrcException("Invalid state");
	fclose(file); file = nullptr;
}

bool FastqFile::ReadString(std::string& str_)
{
	str_.clear();
	int c;
	while ((c = getc_unlocked(file)) != EOF && c != '\n') str_.push_back((char)c);
	return !str_.empty();
}

bool FastqFile::ReadNextRecord(FastqRecord& rec_)
{
	if (!file || writing) throw DsrcException("Invalid state");
	return ReadString(rec_.tag) && ReadString(rec_.sequence) && ReadString(rec_.plus) && ReadString(rec_.quality);
}

void FastqFile::WriteNextRecord(const FastqRecord& rec_)
{
	if (!file || !writing) throw DsrcException("Invalid state");
	const std::string* part[4] = {&rec_.tag, &rec_.sequence, &rec_.plus, &rec_.quality};
	for (const std::string* p : part) { fwrite(p->data(), 1, p->size(), file); putc_unlocked('\n', file); }
}

// ---- DsrcArchive (write side) -----------------------------------------------------------------------------------
struct DsrcArchive::ArchiveImpl
{
	enum State { StateNone, StateCompression, StateDecompression } state = StateNone;
	dsrcgpu_handle* h = nullptr;
	comp::ArchiveWriter* writer = nullptr;
	// reading
	comp::ArchiveReader* reader = nullptr;
	uint64 nextBlock = 0;
	std::vector<uchar> text; uint64 textPos = 0;                      // decoded text of the current batch
	std::vector<std::pair<uint64, uint64> > spans; size_t span = 0;  // [begin, end) of every block's text in `text`
	uint32 lastWord = 0; bool haveLastWord = false;                  // chunkSize word of the block before the current batch
	comp::CompressionSettings settings;
	fq::FastqDatasetType type;
	uint64 bufferSize = 0;
	// chunks waiting for the GPU: FASTQ text without the last newline + the chunkSize word of the block
	std::vector<std::vector<uchar> > chunks;
	std::vector<uint32_t> chunkSizes;
	std::vector<uchar> cur;          // chunk being filled
	uint64 curPayload = 0;           // title + sequence + quality bytes in `cur` (BlockCompressorExt::ChunkSize)
	uint64 runningSize = 0;          // chunkHeader.chunkSize: never cleared between blocks (src/BlockCompressor.cpp:105-109)
	uint64 pendingBytes = 0;
	void Release()
	{
		if (h) { dsrcgpu_destroy(h); h = nullptr; }
		delete writer; writer = nullptr;
		delete reader; reader = nullptr;
		text.clear(); spans.clear(); span = 0; textPos = 0; nextBlock = 0; haveLastWord = false;
		chunks.clear(); chunkSizes.clear(); cur.clear(); curPayload = runningSize = pendingBytes = 0;
		state = StateNone;
	}
};

DsrcArchive::DsrcArchive() : impl(new ArchiveImpl()) { params.qualityOffset = 0; }
DsrcArchive::~DsrcArchive() { impl->Release(); delete impl; }

void DsrcArchive::StartCompress(const std::string& filename_)
{
	if (impl->state != ArchiveImpl::StateNone) throw DsrcException("Invalid state");
	// ArchiveSettings::FromInputParams (src/DsrcArchive.cpp:33-47)
	impl->settings = comp::CompressionSettings();
	impl->settings.dnaOrder = params.dnaCompressionLevel * 3;
	impl->settings.qualityOrder = params.qualityCompressionLevel * 3;
	impl->settings.lossy = params.lossyCompression;
	impl->type.qualityOffset = params.qualityOffset;
	impl->type.plusRepetition = plusRepetition;
	impl->type.colorSpace = colorSpace;
	impl->bufferSize = (uint64)params.fastqBufferSizeMB << 20;
	if (colorSpace) throw DsrcException("DsrcArchive: colour-space records are not supported by the record-level API on the GPU path (use DsrcModule::Compress)");
	// (lossless quality levels 1 and 2 arrive as qualityOrder 3 and 6, as in the reference: the order-2 models without the "F" schemes,
	// the order byte in the footer as it is; the reference's DsrcArchive reads them back, tests/golden/make_records_golden.py)
	if (impl->type.qualityOffset == 0)
		throw DsrcException("DsrcArchive: set the quality offset (33 or 64); the archive API does not analyse the data");
	comp::InputParameters args = params;
	impl->h = comp::DsrcCompressorGPU::CreateInstance(args, impl->settings, impl->type);
	try
	{
		impl->writer = new comp::ArchiveWriter();
		impl->writer->Start(filename_);
	}
	catch (...) { impl->Release(); throw; }
	impl->state = ArchiveImpl::StateCompression;
}

void DsrcArchive::WriteNextRecord(const FastqRecord& rec_)
{
	if (impl->state != ArchiveImpl::StateCompression) throw DsrcException("Invalid state");
	const std::string* part[4] = {&rec_.tag, &rec_.sequence, &rec_.plus, &rec_.quality};
	// what the reference leaves undefined is refused here
	if (rec_.tag.empty() || rec_.tag[0] != '@' || rec_.plus.empty() || rec_.plus[0] != '+' || rec_.sequence.empty()
		|| rec_.sequence.size() != rec_.quality.size() || rec_.sequence.size() > 65535 || rec_.tag.size() > 65535)
		throw DsrcException("DsrcArchive::WriteNextRecord: malformed record (tag '@...', plus '+...', sequence and quality of equal non-zero length)");
	for (const std::string* p : part)
		if (p->find('\n') != std::string::npos || p->find('\r') != std::string::npos)
			throw DsrcException("DsrcArchive::WriteNextRecord: line terminators inside a record string");
	std::vector<uchar>& c = impl->cur;
	for (const std::string* p : part) { c.insert(c.end(), p->begin(), p->end()); c.push_back('\n'); }
	impl->curPayload += rec_.tag.size() + rec_.sequence.size() + rec_.quality.size();
	impl->runningSize += rec_.tag.size() + rec_.sequence.size() + rec_.plus.size() + rec_.quality.size() + 4;    // BlockCompressorExt::RecordSize
	if (impl->curPayload > impl->bufferSize) CloseChunk();
}

void DsrcArchive::CloseChunk()
{
	impl->cur.pop_back();                                    // Store()'s chunks carry no final newline
	impl->pendingBytes += impl->cur.size();
	impl->chunks.emplace_back(); impl->chunks.back().swap(impl->cur);
	impl->chunkSizes.push_back((uint32_t)impl->runningSize);
	impl->curPayload = 0;
	if (impl->pendingBytes >= (1536ull << 20)) FlushBatch();
}

void DsrcArchive::FlushBatch()
{
	const uint32 n = (uint32)impl->chunks.size();
	if (n == 0) return;
	std::vector<const uint8_t*> ptrs(n); std::vector<uint64_t> sizes(n), offs(n), osz(n), raw(4 * n), comp(4 * n);
	uint64 cap = 0;
	for (uint32 i = 0; i < n; ++i) { ptrs[i] = impl->chunks[i].data(); sizes[i] = impl->chunks[i].size(); cap += sizes[i] + (1u << 16); }
	std::vector<uchar> out(cap);
	if (dsrcgpu_set_record_layout(impl->h, n, impl->chunkSizes.data()) != DSRCGPU_OK
		|| dsrcgpu_compress_batch(impl->h, n, ptrs.data(), sizes.data(), out.data(), cap, offs.data(), osz.data(), raw.data(), comp.data()) != DSRCGPU_OK)
	{
		const std::string msg = dsrcgpu_last_error(impl->h);
		impl->Release();
		throw DsrcException(msg);
	}
	for (uint32 i = 0; i < n; ++i) impl->writer->WriteBlock(out.data() + offs[i], osz[i], (const uint64*)&raw[4 * i], (const uint64*)&comp[4 * i]);
	impl->chunks.clear(); impl->chunkSizes.clear(); impl->pendingBytes = 0;
}

void DsrcArchive::FinishCompress()
{
	if (impl->state != ArchiveImpl::StateCompression) throw DsrcException("Invalid state");
	if (impl->curPayload > 0) CloseChunk();
	FlushBatch();
	impl->writer->Finish(impl->type, impl->settings);
	impl->Release();
}

void DsrcArchive::StartDecompress(const std::string& filename_)
{
	if (impl->state != ArchiveImpl::StateNone) throw DsrcException("Invalid state");
	impl->reader = new comp::ArchiveReader();
	try
	{
		impl->reader->Open(filename_);
		// ArchiveSettings::ToInputParams (src/DsrcArchive.cpp:49-63): the archive's settings become the object's
		impl->settings = impl->reader->Settings(); impl->type = impl->reader->Type();
		params.dnaCompressionLevel = impl->settings.dnaOrder / 3;
		params.qualityCompressionLevel = impl->settings.qualityOrder / 3;
		params.lossyCompression = impl->settings.lossy;
		params.calculateCrc32 = impl->settings.calculateCrc32;
		params.tagPreserveFlags = impl->settings.tagPreserveFlags;
		params.qualityOffset = impl->type.qualityOffset;
		plusRepetition = impl->type.plusRepetition; colorSpace = impl->type.colorSpace;
		impl->h = comp::CreateDecodeInstance(params.device, impl->settings, impl->type);
	}
	catch (...) { impl->Release(); throw; }
	impl->nextBlock = 0; impl->text.clear(); impl->textPos = 0;
	impl->state = ArchiveImpl::StateDecompression;
}

// BlockCompressorExt::Feed for the next batch of blocks (src/BlockCompressorExt.cpp:59-66)
bool DsrcArchive::FeedBatch()
{
	comp::ArchiveReader& rd = *impl->reader;
	const uint64 nBlocks = rd.BlockCount();
	if (impl->nextBlock >= nBlocks) return false;
	uint64 hi = impl->nextBlock, bytes = 0;
	while (hi < nBlocks && (hi == impl->nextBlock || bytes + rd.BlockSizes()[hi] <= (64ull << 20))) { bytes += rd.BlockSizes()[hi]; ++hi; }
	const uint32 n = (uint32)(hi - impl->nextBlock);
	std::vector<std::vector<uchar> > blocks(n);
	std::vector<const uint8_t*> ptrs(n); std::vector<uint64_t> sizes(n), offs(n), tsz(n), caps; std::vector<uint32> words(n);
	for (uint32 i = 0; i < n; ++i)
	{
		sizes[i] = rd.BlockSizes()[impl->nextBlock + i];
		if (sizes[i] < 16) throw DsrcException("Corrupted DSRC archive: block too small");
		blocks[i].resize(sizes[i]); rd.ReadBlock(impl->nextBlock + i, blocks[i].data()); ptrs[i] = blocks[i].data();
		words[i] = (uint32)comp::GetBE(ptrs[i] + 12, 4);
	}
	int rc = DSRCGPU_OK;
	for (int attempt = 0; attempt < 2; ++attempt)
	{
		comp::TextCaps(words, sizes, impl->lastWord, impl->haveLastWord, attempt == 0, caps);
		uint64 cap = 0; for (uint64 c : caps) cap += c;
		impl->text.resize(cap + 64);
		rc = dsrcgpu_decompress_batch(impl->h, n, ptrs.data(), sizes.data(), caps.data(), impl->text.data(), impl->text.size(), offs.data(), tsz.data(), nullptr);
		if (rc != DSRCGPU_E_CAPACITY) break;
	}
	if (rc != DSRCGPU_OK) { const std::string msg = dsrcgpu_last_error(impl->h); throw DsrcException(msg); }
	impl->lastWord = words[n - 1]; impl->haveLastWord = true;
	impl->spans.clear();
	for (uint32 i = 0; i < n; ++i) impl->spans.emplace_back(offs[i], offs[i] + tsz[i]);
	impl->span = 0; impl->textPos = n ? offs[0] : 0;
	impl->nextBlock = hi;
	return true;
}

// BlockCompressorExt::ExtractNextRecord (src/BlockCompressorExt.cpp:128-147): tag / sequence / quality as decoded; plus is
// the tag with '+' in front when the archive repeats titles, else "+"
bool DsrcArchive::ReadNextRecord(FastqRecord& rec_)
{
	if (impl->state != ArchiveImpl::StateDecompression) throw DsrcException("Invalid state");
	for (;;)
	{
		while (impl->span < impl->spans.size() && impl->textPos >= impl->spans[impl->span].second)
		{
			++impl->span;
			if (impl->span < impl->spans.size()) impl->textPos = impl->spans[impl->span].first;
		}
		if (impl->span < impl->spans.size()) break;
		if (!FeedBatch()) return false;
	}
	const uchar* t = impl->text.data(); const uint64 end = impl->spans[impl->span].second;
	std::string* part[4] = {&rec_.tag, &rec_.sequence, &rec_.plus, &rec_.quality};
	for (int k = 0; k < 4; ++k)
	{
		uint64 e = impl->textPos;
		while (e < end && t[e] != '\n') ++e;
		if (k != 2) part[k]->assign((const char*)t + impl->textPos, (size_t)(e - impl->textPos));
		impl->textPos = e < end ? e + 1 : e;
	}
	if (impl->type.plusRepetition) { rec_.plus = rec_.tag; if (!rec_.plus.empty()) rec_.plus[0] = '+'; }
	else if (rec_.plus.length() != 1) rec_.plus.assign(1, '+');
	return true;
}

void DsrcArchive::FinishDecompress()
{
	if (impl->state != ArchiveImpl::StateDecompression) throw DsrcException("Invalid state");
	impl->Release();
}

void DsrcModule::Decompress(const std::string& in, const std::string& out)
{
	comp::DsrcDecompressorGPU op;
	comp::InputParameters p = params;
	p.inputFilename = in; p.outputFilename = out;
	if (!op.Process(p)) throw DsrcException(op.GetError());
}
} // namespace wrap
} // namespace dsrc
