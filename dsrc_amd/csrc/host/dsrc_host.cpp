// Host side of the MI355X DSRC compressor (see dsrc_host.h).  Plain C++17, links libdsrc_gpu.so.
#include "dsrc_host.h"

#include <algorithm>
#include <cstring>
#include <iomanip>
#include <sstream>

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
	f = fopen(path.c_str(), "wb");
	if (!f) throw DsrcException("Cannot open file to write:" + path);
	uchar zero[40]; memset(zero, 0, sizeof(zero));
	fwrite(zero, 1, 40, f);                          // header is written last (src/DsrcFile.cpp:52-54)
}

void ArchiveWriter::WriteBlock(const uchar* data, uint64 size, const uint64 raw[4], const uint64 comp[4])
{
	fwrite(data, 1, size, f);
	blockSizes.push_back((uint32)size);
	for (int i = 0; i < 4; ++i) { rawInfo.sizes[i] += raw[i]; compInfo.sizes[i] += comp[i]; }
}

void ArchiveWriter::Finish(const fq::FastqDatasetType& type, const CompressionSettings& s)
{
	const uint64 footerOffset = (uint64)ftello(f);
	std::vector<uchar> foot;
	foot.push_back(0xCC);
	const uchar* bs = (const uchar*)blockSizes.data();     // host-endian uint32 array, as the reference writes it
	foot.insert(foot.end(), bs, bs + blockSizes.size() * 4);
	foot.push_back((uchar)((type.colorSpace ? 2 : 0) | (type.plusRepetition ? 1 : 0)));
	foot.push_back((uchar)type.qualityOffset);
	foot.push_back((uchar)((s.lossy ? 1 : 0) | (s.calculateCrc32 ? 2 : 0)));
	foot.push_back((uchar)s.dnaOrder); foot.push_back((uchar)s.qualityOrder);
	PutBE(foot, s.tagPreserveFlags, 8);
	fwrite(foot.data(), 1, foot.size(), f);
	std::vector<uchar> head;
	head.push_back(0xAA); head.push_back(2); head.push_back(0); head.push_back(2);
	PutBE(head, foot.size(), 4); PutBE(head, footerOffset, 8); PutBE(head, 0, 8); PutBE(head, blockSizes.size(), 8);
	for (int i = 0; i < 8; ++i) head.push_back(0xAA);
	fseeko(f, 0, SEEK_SET);
	fwrite(head.data(), 1, head.size(), f);
	fclose(f); f = nullptr;
}

ArchiveWriter::~ArchiveWriter() { if (f) fclose(f); }

// ---- operator -----------------------------------------------------------------------------------------------
bool DsrcCompressorGPU::Process(const InputParameters& args)
{
	dsrcgpu_handle* h = nullptr;
	FILE* in = nullptr;
	try
	{
		in = args.useFastqStdIo ? stdin : fopen(args.inputFilename.c_str(), "rb");
		if (!in) throw DsrcException("Cannot open file to read:" + args.inputFilename);
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

		dsrcgpu_settings gs; memset(&gs, 0, sizeof(gs));
		gs.dna_order = settings.dnaOrder; gs.quality_order = settings.qualityOrder; gs.tag_preserve_flags = settings.tagPreserveFlags;
		gs.lossy = settings.lossy; gs.calculate_crc32 = settings.calculateCrc32;
		dsrcgpu_dataset gd; memset(&gd, 0, sizeof(gd));
		gd.quality_offset = type.qualityOffset; gd.plus_repetition = type.plusRepetition; gd.color_space = type.colorSpace;
		if (dsrcgpu_create(&gs, &gd, args.device, 0, &h) != DSRCGPU_OK)
			throw DsrcException(std::string(h ? dsrcgpu_last_error(h) : "cannot create the GPU compressor"));

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
			for (uint32 i = 0; i < n; ++i)
			{
				uint64 r4[4], c4[4];
				for (int k = 0; k < 4; ++k) { r4[k] = raw[4 * i + k]; c4[k] = comp[4 * i + k]; }
				writer.WriteBlock(out.data() + offs[i], osz[i], r4, c4);
			}
			chunks.clear();
		}
		writer.Finish(type, settings);

		std::ostringstream ss;          // same text as the reference's -v log (src/DsrcOperator.cpp:362-375)
		const fq::StreamsInfo& rawS = writer.Raw(); const fq::StreamsInfo& compS = writer.Comp();
		ss << "Compressed streams sizes (in bytes)\n";
		ss << "TAG: " << std::setw(16) << compS.sizes[fq::StreamsInfo::MetaStream] + compS.sizes[fq::StreamsInfo::TagStream]
		   << " / " << std::setw(16) << rawS.sizes[fq::StreamsInfo::TagStream] << '\n';
		ss << "DNA: " << std::setw(16) << compS.sizes[fq::StreamsInfo::DnaStream] << " / " << std::setw(16) << rawS.sizes[fq::StreamsInfo::DnaStream] << '\n';
		ss << "QUA: " << std::setw(16) << compS.sizes[fq::StreamsInfo::QualityStream] << " / " << std::setw(16) << rawS.sizes[fq::StreamsInfo::QualityStream] << '\n';
		AddLog(ss.str());
	}
	catch (const DsrcException& e) { AddError(e.what()); }
	catch (const std::exception& e) { AddError(e.what()); }
	if (h) dsrcgpu_destroy(h);
	if (in && in != stdin) fclose(in);
	return !IsError();
}

} // namespace comp

namespace wrap
{
// range checks of the reference's setters (src/Configurable.cpp:56-179)
void Configurable::SetFastqBufferSizeMB(uint64 s) { if (s < 1 || s > 1024) throw DsrcException("Invalid fastq buffer size specified [1-1024]"); params.fastqBufferSizeMB = (uint32)s; }
void Configurable::SetDnaCompressionLevel(uint32 l) { if (l > 3) throw DsrcException("Invalid DNA compression mode specified [0-3]"); params.dnaCompressionLevel = l; }
void Configurable::SetQualityCompressionLevel(uint32 l) { if (l > 2) throw DsrcException("Invalid Quality compression mode specified [0-2]"); params.qualityCompressionLevel = l; }
void Configurable::SetQualityOffset(uint32 o) { if (o != 0 && (o < 33 || o > 64)) throw DsrcException("Invalid Quality offset mode specified [33, 64]"); params.qualityOffset = o; }
void Configurable::SetThreadsNumber(uint32 t) { if (t == 0 || t > 64) throw DsrcException("Invalid thread number specified [1-64]"); params.threadNum = t; }

void DsrcModule::Compress(const std::string& in, const std::string& out)
{
	comp::DsrcCompressorGPU op;
	comp::InputParameters p = params;
	p.inputFilename = in; p.outputFilename = out;
	if (!op.Process(p)) throw DsrcException(op.GetError());
}

void DsrcModule::Decompress(const std::string&, const std::string&)
{
	throw DsrcException("Decompression is not part of the MI355X hot path (SURVEY 8f-1); use the reference's DsrcModule::Decompress");
}
} // namespace wrap
} // namespace dsrc
