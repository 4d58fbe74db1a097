// Host side of the MI355X DSRC compressor, above the C ABI (include/dsrc_gpu.h).
//
// Mirrors the reference's operator / module interface for the compression path so that callers of
//   dsrc::comp::IDsrcOperator::Process(const InputParameters&)      (src/DsrcOperator.h:26-91)
//   dsrc::wrap::DsrcModule::Compress(in, out) + Configurable setters  (include/dsrc/DsrcModule.h:22-40,
//                                                                     include/dsrc/Configurable.h:51-82)
// keep their code: same names, same argument meaning, same error convention (Process returns false and
// GetError() carries "Error: ...\n"; DsrcModule throws DsrcException).  What changes is below Process():
// the reader thread + N DsrcCompressor workers + writer become
//   FastqChunker (IFastqStreamReader::ReadNextChunk, src/FastqStream.cpp:18-98)
//   -> batches of chunks -> dsrcgpu_compress_batch (GPU block scheduler)
//   -> ArchiveWriter (DsrcFileWriter, src/DsrcFile.cpp:38-170).
#pragma once

#include <cstdint>
#include <cstdio>
#include <exception>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

struct dsrcgpu_handle;

namespace dsrc
{

typedef unsigned char uchar, byte;
typedef unsigned int uint32;
typedef unsigned long long uint64;
typedef long long int64;

class DsrcException : public std::exception
{
	std::string message;
public:
	explicit DsrcException(const std::string& m) : message(m) {}
	~DsrcException() throw() {}
	const char* what() const throw() { return message.c_str(); }
};

namespace fq
{
struct FastqDatasetType
{
	static const uint32 AutoQualityOffset = 0;
	uint32 qualityOffset = AutoQualityOffset;
	bool plusRepetition = false;
	bool colorSpace = false;
};

struct StreamsInfo
{
	enum StreamName { MetaStream = 0, TagStream, DnaStream, QualityStream, StreamCount = 4 };
	uint64 sizes[4] = {0, 0, 0, 0};
};
} // namespace fq

namespace comp
{

struct CompressionSettings
{
	uint32 dnaOrder = 0;
	uint32 qualityOrder = 0;
	uint64 tagPreserveFlags = 0;
	bool lossy = false;
	bool calculateCrc32 = false;
};

struct InputParameters          // src/Common.h:149-193, plus the GPU knobs at the end
{
	uint32 qualityOffset = 0;
	uint32 dnaCompressionLevel = 0;
	uint32 qualityCompressionLevel = 0;
	uint32 threadNum = 4;          // here: GPU scheduler instances (host threads) working on consecutive batches, 1..8
	uint64 tagPreserveFlags = 0;
	uint32 fastqBufferSizeMB = 8;
	bool lossyCompression = false;
	bool calculateCrc32 = false;
	bool verifyCrc32 = true;        // with calculateCrc32: decode every block after writing it and compare (reference behaviour, src/DsrcWorker.cpp:53-62)
	bool useFastqStdIo = false;
	std::string inputFilename;
	std::string outputFilename;
	// GPU path
	int device = 0;
	std::vector<int> devices;      // more than one: scheduler instances are spread over these GPUs of the node (threadNum instances EACH); one archive
	uint32 batchBlocks = 0;        // chunks per scheduler pass; 0 = as many as fit ~1.5 GiB of input
	bool exitWhenDone = false;     // command line only: leave the process (_exit) as soon as the output file is complete
	bool verboseLog = false;       // with exitWhenDone: print the log before leaving
};

class IDsrcOperator
{
public:
	virtual ~IDsrcOperator() {}
	virtual bool Process(const InputParameters& args_) = 0;
	bool IsError() const { return errorMessage.length() > 0; }
	const std::string& GetError() const { return errorMessage; }
	void ClearError() { errorMessage.clear(); }
	const std::string& GetLog() const { return logMessage; }
	void ClearLog() { logMessage.clear(); }

	// level -> order mapping of the reference (src/DsrcOperator.h:74-90)
	static CompressionSettings GetCompressionSettings(const InputParameters& args_);

protected:
	std::string errorMessage, logMessage;
	void AddError(const std::string& e) { errorMessage += "Error: " + e + '\n'; }
	void AddLog(const std::string& l) { logMessage += l + '\n'; }
};

// Drop-in for DsrcCompressorMT / DsrcCompressorST (src/DsrcOperator.cpp:55-395)
class ArchiveWriter;
class DsrcCompressorGPU : public IDsrcOperator
{
public:
	bool Process(const InputParameters& args_);
	static dsrcgpu_handle* CreateInstance(const InputParameters& args_, const CompressionSettings& settings_, const fq::FastqDatasetType& type_, int device_ = -1);
private:
	bool ProcessStream(const InputParameters& args_, FILE* in_);      // stdin / pipes: one scheduler instance
	void LogSizes(const ArchiveWriter& writer_);
};

// Drop-in for DsrcDecompressorMT / DsrcDecompressorST (src/DsrcOperator.cpp:397-521): DsrcFileReader -> batches of
// blocks -> dsrcgpu_decompress_batch (GPU block scheduler) -> FASTQ file / stdout, block order kept.
class DsrcDecompressorGPU : public IDsrcOperator
{
public:
	bool Process(const InputParameters& args_);
};

// ---- pieces (exposed for tests) -----------------------------------------------------------------------------

// IFastqStreamReader::ReadNextChunk on a FILE* (src/FastqStream.cpp:18-98)
class FastqChunker
{
public:
	FastqChunker(FILE* f, uint64 bufferSize);
	// returns false at end of input; chunk.size() is FastqDataChunk::size (final newline not included)
	bool ReadNextChunk(std::vector<uchar>& chunk);
	static uint64 NextRecordPos(const uchar* d, uint64 pos, uint64 size, bool& crlf);
private:
	FILE* file;
	uint64 bufSize;
	std::vector<uchar> carry;
	bool eof = false, usesCrlf = false;
};

// FastqParser::Analyze (src/FastqParser.cpp:27-138)
bool AnalyzeFirstChunk(const uchar* data, uint64 size, fq::FastqDatasetType& type, bool estimateQualityOffset);

// DsrcFileWriter (src/DsrcFile.cpp:38-170)
class ArchiveWriter
{
public:
	void Start(const std::string& path);
	// Regular files: a side thread reserves the file's space ahead of the writers (fallocate, 256 MiB at a time, up to `expect` bytes;
	// Finish cuts the file back to what was written).  On tmpfs one fallocate call hands out pages at 18 GB/s while writes that have
	// to allocate their pages manage 4-7 GB/s for the whole file however many threads issue them (the inode's lock): with the pages in
	// place the same writes reach 9-11 GB/s (profiles/r05_tmpfs_probe.txt).  A file system without fallocate: nothing happens.
	void ReserveAhead(uint64 expect);
	void WriteBlock(const uchar* data, uint64 size, const uint64 raw[4], const uint64 comp[4]);
	// for writers that run in parallel: Claim (one caller at a time, in archive order) reserves the file range of the next
	// n blocks and records their sizes; WriteAt (any thread) fills it
	uint64 Claim(uint32 n, const uint64_t* sizes, const uint64_t* raw, const uint64_t* comp);
	void WriteAt(uint64 off, const void* p, uint64 n) const;
	void Finish(const fq::FastqDatasetType& type, const CompressionSettings& settings);
	const fq::StreamsInfo& Raw() const { return rawInfo; }
	const fq::StreamsInfo& Comp() const { return compInfo; }
	void Abandon();                  // close and remove the unfinished archive
	~ArchiveWriter();
private:
	std::string name;
	int fd = -1;
	uint64 pos = 0;
	std::vector<uint32> blockSizes;
	fq::StreamsInfo rawInfo, compInfo;
	std::thread reserver; std::atomic<bool> reserveStop{false}; std::atomic<uint64> reserved{0};
	void StopReserver();
};

// DsrcFileReader (src/DsrcFile.cpp:172-318): header, footer (block sizes, dataset type, compression settings)
class ArchiveReader
{
public:
	~ArchiveReader();
	void Open(const std::string& path);           // throws DsrcException with the reference's messages
	void Close();
	int Fd() const { return fd; }
	uint64 BlockCount() const { return blockSizes.size(); }
	const std::vector<uint32>& BlockSizes() const { return blockSizes; }
	uint64 BlockOffset(uint64 i) const { return blockOffs[i]; }
	const fq::FastqDatasetType& Type() const { return type; }
	const CompressionSettings& Settings() const { return settings; }
	void ReadBlock(uint64 i, uchar* dst) const;
private:
	int fd = -1;
	std::vector<uint32> blockSizes;
	std::vector<uint64> blockOffs;
	fq::FastqDatasetType type;
	CompressionSettings settings;
};

// helpers shared with the record-level API
uint64 GetBE(const uchar* p, int bytes);
void TextCaps(const std::vector<uint32>& words, const std::vector<uint64_t>& blockSizes, uint32 wordBefore, bool haveBefore, bool exact, std::vector<uint64_t>& caps);
dsrcgpu_handle* CreateDecodeInstance(int device, const CompressionSettings& settings, const fq::FastqDatasetType& type);

} // namespace comp

namespace wrap
{

class Configurable           // include/dsrc/Configurable.h:51-82 (compression-side setters)
{
public:
	void SetFastqBufferSizeMB(uint64 size_);
	uint64 GetFastqBufferSizeMB() const { return params.fastqBufferSizeMB; }
	void SetDnaCompressionLevel(uint32 level_);
	uint32 GetDnaCompressionLevel() const { return params.dnaCompressionLevel; }
	void SetQualityCompressionLevel(uint32 level_);
	uint32 GetQualityCompressionLevel() const { return params.qualityCompressionLevel; }
	void SetLossyCompression(bool lossy_) { params.lossyCompression = lossy_; }
	bool IsLossyCompression() const { return params.lossyCompression; }
	void SetQualityOffset(uint32 off_);
	uint32 GetQualityOffset() const { return params.qualityOffset; }
	void SetThreadsNumber(uint32 threadNum_);
	uint32 GetThreadsNumber() const { return params.threadNum; }
	void SetStdIoUsing(bool use_) { params.useFastqStdIo = use_; }
	bool IsStdIoUsing() const { return params.useFastqStdIo; }
	void SetCrc32Checking(bool use_) { params.calculateCrc32 = use_; }
	bool IsCrc32Checking() const { return params.calculateCrc32; }
	void SetTagFieldFilterMask(uint64 mask_) { params.tagPreserveFlags = mask_; }
	uint64 GetTagFieldFilterMask() const { return params.tagPreserveFlags; }
	void SetPlusRepetition(bool use_) { plusRepetition = use_; }
	bool IsPlusRepetition() const { return plusRepetition; }
	void SetColorSpace(bool use_) { colorSpace = use_; }
	bool IsColorSpace() const { return colorSpace; }
	void SetDevice(int device_) { params.device = device_; }
	int GetDevice() const { return params.device; }
protected:
	comp::InputParameters params;
	bool plusRepetition = false, colorSpace = false;
};

class FieldMask              // include/dsrc/Configurable.h:22-43
{
public:
	FieldMask() : mask(0) {}
	FieldMask AddField(uint32 i_) const { FieldMask m(*this); m.mask |= 1ull << i_; return m; }
	uint64 GetMask() const { return mask; }
private:
	uint64 mask;
};

struct FastqRecord           // include/dsrc/FastqRecord.h:21-27
{
	std::string tag, sequence, plus, quality;
};

// include/dsrc/FastqFile.h:22-85, src/FastqFile.cpp: strings up to '\n'; an empty string ends the file
class FastqFile
{
public:
	FastqFile() {}
	~FastqFile();
	void Open(const std::string& filename_);
	void Create(const std::string& filename_);
	void Close();
	bool ReadNextRecord(FastqRecord& rec_);
	void WriteNextRecord(const FastqRecord& rec_);
private:
	FILE* file = nullptr;
	bool writing = false;
	bool ReadString(std::string& str_);
	FastqFile(const FastqFile&); FastqFile& operator=(const FastqFile&);
};

// Record-level archive API, write side (include/dsrc/DsrcArchive.h:27-66, src/DsrcArchive.cpp:100-150,217-224,
// src/BlockCompressorExt.cpp).  Records are gathered into chunks exactly as BlockCompressorExt does (a chunk is closed
// once its title+sequence+quality bytes exceed FastqBufferSizeMB), chunks are compressed on the GPU in batches through
// dsrcgpu_set_record_layout + dsrcgpu_compress_batch, and the archive is the one the reference's DsrcArchive writes.
// As in the reference this API maps QualityCompressionLevel to qualityOrder = 3 * level whether or not the mode is
// lossy, so lossless archives are only defined for level 0 (others are refused), and it ignores Crc32Checking and
// TagFieldFilterMask.  Reading (StartDecompress / ReadNextRecord / FinishDecompress, src/DsrcArchive.cpp:170-215,
// BlockCompressorExt::Feed / ExtractNextRecord src/BlockCompressorExt.cpp:49-66,128-147): blocks are decoded on the GPU a
// batch at a time and handed out record by record; the settings come from the archive's footer.
class DsrcArchive : public Configurable
{
public:
	DsrcArchive();
	~DsrcArchive();
	void StartCompress(const std::string& filename_);
	void WriteNextRecord(const FastqRecord& rec_);
	void FinishCompress();
	void StartDecompress(const std::string& filename_);
	bool ReadNextRecord(FastqRecord& rec_);
	void FinishDecompress();
private:
	struct ArchiveImpl;
	ArchiveImpl* impl;
	void CloseChunk();
	void FlushBatch();
	bool FeedBatch();
	DsrcArchive(const DsrcArchive&); DsrcArchive& operator=(const DsrcArchive&);
};

class DsrcModule : public Configurable   // include/dsrc/DsrcModule.h:22-40
{
public:
	void Compress(const std::string& inputFilename_, const std::string& outputFilename_);
	void Decompress(const std::string& inputFilename_, const std::string& outputFilename_);
};

} // namespace wrap
} // namespace dsrc
