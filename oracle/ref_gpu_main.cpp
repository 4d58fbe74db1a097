// TEST INFRASTRUCTURE ONLY -- never linked into the product.
//
// Proves the drop-in boundary against the REAL reference: the unmodified reference sources (chunk reader, queues,
// pools, ordered archive writer, archive reader, FASTQ writer -- compiled where they lie by oracle/Makefile) with their
// N CPU worker threads replaced by ONE worker that talks to libdsrc_gpu.so through include/dsrc_gpu.h.  The two worker
// classes below are the binding INTEGRATION.md section 1 tells a reference maintainer to add (src/DsrcWorkerGpu.h there);
// main() sets the pipeline up the way DsrcCompressorMT::Process / DsrcDecompressorMT::Process do
// (src/DsrcOperator.cpp:230-521) and starts the GPU worker where those start `threadNum` DsrcCompressor /
// DsrcDecompressor objects.  tests/test_boundary_ref.py checks that the archives are the ones `dsrc_ref c -t1` writes.
//
//   dsrc_ref_gpu c [-d<n>] [-q<n>] [-l] [-c] [-b<MB>] [-n<chunks per batch>] in.fastq out.dsrc
//   dsrc_ref_gpu d [-n<blocks per batch>] in.dsrc out.fastq
//
// This file contains no reference code: it only calls it.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "DsrcOperator.h"
#include "DsrcWorker.h"
#include "DsrcIo.h"
#include "DsrcFile.h"
#include "FastqIo.h"
#include "FastqStream.h"
#include "ErrorHandler.h"

#include "dsrc_gpu.h"

using namespace dsrc;

namespace
{

dsrcgpu_handle* make_handle(const fq::FastqDatasetType& type, const comp::CompressionSettings& cs, core::ErrorHandler& eh, bool verify)
{
	dsrcgpu_settings s; std::memset(&s, 0, sizeof(s));
	s.dna_order = cs.dnaOrder; s.quality_order = cs.qualityOrder; s.tag_preserve_flags = cs.tagPreserveFlags;
	s.lossy = cs.lossy; s.calculate_crc32 = cs.calculateCrc32; s.verify_after_compress = verify && cs.calculateCrc32;
	dsrcgpu_dataset d; std::memset(&d, 0, sizeof(d));
	d.quality_offset = type.qualityOffset; d.plus_repetition = type.plusRepetition; d.color_space = type.colorSpace;
	dsrcgpu_handle* h = NULL;
	if (dsrcgpu_create(&s, &d, 0, 0, &h) != DSRCGPU_OK)
	{
		eh.SetError(h ? dsrcgpu_last_error(h) : "dsrcgpu_create failed");
		if (h) dsrcgpu_destroy(h);
		return NULL;
	}
	return h;
}

// replaces the pool of comp::DsrcCompressor workers (src/DsrcWorker.cpp:30-73)
class DsrcCompressorGpu : public comp::IDsrcThreadWorker
{
public:
	DsrcCompressorGpu(fq::FastqDataQueue& fq_, fq::FastqDataPool& fp_, comp::DsrcDataQueue& dq_, comp::DsrcDataPool& dp_, core::ErrorHandler& eh_,
					  const fq::FastqDatasetType& t_, const comp::CompressionSettings& s_, uint32 batch_, uint32 producers_)
		: comp::IDsrcThreadWorker(fq_, fp_, dq_, dp_, eh_, t_, s_), batchBlocks(batch_), producersExpected(producers_) {}

private:
	uint32 batchBlocks, producersExpected;

	// blocks that are ready go to the reference's writer queue; `wait`: also the ones still being compressed; `one`: a single block
	bool Drain(dsrcgpu_handle* h, bool wait, bool one = false)
	{
		for (bool first = true;; first = false)
		{
			if (one && !first) return true;
			int64_t id; uint8_t* blk; uint64_t size, raw[4], cmp[4];
			const int rc = wait ? dsrcgpu_collect(h, &id, &blk, &size, raw, cmp) : dsrcgpu_try_collect(h, &id, &blk, &size, raw, cmp);
			if (rc == 0) return true;
			if (rc < 0) { errorHandler.SetError(dsrcgpu_last_error(h)); return false; }
			comp::DsrcDataChunk* out = NULL;
			dsrcPool.Acquire(out);
			if (out->data.Size() < size) out->data.Extend(size);
			std::memcpy(out->data.Pointer(), blk, size);
			out->size = size;
			for (int i = 0; i < 4; ++i) { out->rawStreamsInfo.sizes[i] = raw[i]; out->compStreamsInfo.sizes[i] = cmp[i]; }
			dsrcgpu_release(h, blk);
			dsrcQueue.Push(id, out);                       // DsrcWriter restores partId order (src/DsrcIo.cpp:25-66)
		}
	}

	void Process()
	{
		dsrcgpu_handle* h = make_handle(datasetType, compSettings, errorHandler, true);
		int64 partId = 0; fq::FastqDataChunk* chunk = NULL; uint32 pending = 0;
		bool more = h != NULL;
		while (more && !errorHandler.IsError())
		{
			more = fastqQueue.Pop(partId, chunk);          // the queue the CPU workers pop from
			if (more)
			{
				int rc;
				// ring full: this thread is also the collector -- hand finished blocks to the writer until a slot is free
				while ((rc = dsrcgpu_submit(h, partId, chunk->data.Pointer(), chunk->size)) == DSRCGPU_E_BUSY)
					if (!Drain(h, true, true)) break;
				if (rc != DSRCGPU_OK) { if (!errorHandler.IsError()) errorHandler.SetError(dsrcgpu_last_error(h)); break; }
				fastqPool.Release(chunk); chunk = NULL; ++pending;
			}
			if (pending == batchBlocks || (!more && pending))
			{
				if (dsrcgpu_flush(h) != DSRCGPU_OK) { errorHandler.SetError(dsrcgpu_last_error(h)); break; }
				pending = 0;
				if (!Drain(h, false)) break;               // whatever earlier batches have finished; the new one runs on
			}
		}
		if (h && !errorHandler.IsError()) Drain(h, true);
		if (h) dsrcgpu_destroy(h);
		// with an error the reader may still be blocked on a full queue / empty pool: keep taking its chunks
		while (errorHandler.IsError() && fastqQueue.Pop(partId, chunk)) fastqPool.Release(chunk);
		for (uint32 i = 0; i < producersExpected; ++i) dsrcQueue.SetCompleted();      // the queue waits for that many (src/DataQueue.h:79-87)
	}
};

// replaces the pool of comp::DsrcDecompressor workers (src/DsrcWorker.cpp:75-104)
class DsrcDecompressorGpu : public comp::IDsrcThreadWorker
{
public:
	DsrcDecompressorGpu(fq::FastqDataQueue& fq_, fq::FastqDataPool& fp_, comp::DsrcDataQueue& dq_, comp::DsrcDataPool& dp_, core::ErrorHandler& eh_,
						const fq::FastqDatasetType& t_, const comp::CompressionSettings& s_, uint32 batch_, uint32 producers_)
		: comp::IDsrcThreadWorker(fq_, fp_, dq_, dp_, eh_, t_, s_), batchBlocks(batch_), producersExpected(producers_) {}

private:
	uint32 batchBlocks, producersExpected;

	void Process()
	{
		dsrcgpu_handle* h = make_handle(datasetType, compSettings, errorHandler, false);
		std::vector<int64> ids; std::vector<comp::DsrcDataChunk*> blocks;
		bool more = h != NULL;
		while (more && !errorHandler.IsError())
		{
			int64 partId = 0; comp::DsrcDataChunk* data = NULL;
			more = dsrcQueue.Pop(partId, data);
			if (more) { ids.push_back(partId); blocks.push_back(data); }
			if (blocks.size() == batchBlocks || (!more && !blocks.empty()))
			{
				const uint32 n = (uint32)blocks.size();
				std::vector<const uint8_t*> ptrs(n); std::vector<uint64_t> sizes(n), offs(n), tsz(n);
				uint64 cap = 0;
				for (uint32 i = 0; i < n; ++i)
				{
					ptrs[i] = blocks[i]->data.Pointer(); sizes[i] = blocks[i]->size;
					const uint8_t* p = ptrs[i];
					cap += ((uint64)p[12] << 24 | (uint64)p[13] << 16 | (uint64)p[14] << 8 | p[15]) + 1;      // chunkSize word + the last newline
				}
				std::vector<uint8_t> text(cap + 64);
				if (dsrcgpu_decompress_batch(h, n, ptrs.data(), sizes.data(), NULL, text.data(), text.size(), offs.data(), tsz.data(), NULL) != DSRCGPU_OK)
					{ errorHandler.SetError(dsrcgpu_last_error(h)); break; }
				for (uint32 i = 0; i < n; ++i)
				{
					fq::FastqDataChunk* out = NULL;
					fastqPool.Acquire(out);
					if (out->data.Size() < tsz[i]) out->data.Extend(tsz[i]);
					std::memcpy(out->data.Pointer(), text.data() + offs[i], tsz[i]);
					out->size = tsz[i];
					fastqQueue.Push(ids[i], out);          // FastqWriter restores partId order (src/FastqIo.cpp:71-135)
					dsrcPool.Release(blocks[i]);
				}
				ids.clear(); blocks.clear();
			}
		}
		for (size_t i = 0; i < blocks.size(); ++i) dsrcPool.Release(blocks[i]);
		if (h) dsrcgpu_destroy(h);
		{ int64 id; comp::DsrcDataChunk* d = NULL; while (errorHandler.IsError() && dsrcQueue.Pop(id, d)) dsrcPool.Release(d); }
		for (uint32 i = 0; i < producersExpected; ++i) fastqQueue.SetCompleted();
	}
};

// level -> order mapping of the reference (a protected static of IDsrcOperator, src/DsrcOperator.h:74-90)
struct LevelMap : public comp::IDsrcOperator
{
	bool Process(const comp::InputParameters&) { return false; }
	static comp::CompressionSettings Of(const comp::InputParameters& a) { return GetCompressionSettings(a); }
};

int compress(const comp::InputParameters& args, uint32 batch)
{
	const comp::CompressionSettings settings = LevelMap::Of(args);
	fq::FastqDatasetType type;
	const uint32 threads = 1, parts = 8;
	fq::FastqFileReader reader(args.inputFilename);
	comp::DsrcFileWriter writer;
	writer.StartCompress(args.outputFilename);
	fq::FastqDataPool fastqPool(parts, (uint64)args.fastqBufferSizeMB << 20);
	fq::FastqDataQueue fastqQueue(parts, 1);
	comp::DsrcDataPool dsrcPool(parts * 64, (uint64)args.fastqBufferSizeMB << 20);         // a batch of blocks is pushed at once
	comp::DsrcDataQueue dsrcQueue(parts * 64, threads);
	core::MultithreadedErrorHandler* eh = new core::MultithreadedErrorHandler();      // value-initialised, as the reference creates it
	core::ErrorHandler& errors = *eh;
	fq::FastqReader dataReader(reader, fastqQueue, fastqPool, errors);
	comp::DsrcWriter dataWriter(writer, dsrcQueue, dsrcPool, errors);
	const bool findOffset = args.qualityOffset == fq::FastqDatasetType::AutoQualityOffset;
	if (!findOffset) type.qualityOffset = args.qualityOffset;
	if (!dataReader.AnalyzeFirstChunk(type, findOffset)) { std::fprintf(stderr, "Error analyzing FASTQ dataset\n"); return 1; }
	writer.SetDatasetType(type);
	writer.SetCompressionSettings(settings);

	DsrcCompressorGpu worker(fastqQueue, fastqPool, dsrcQueue, dsrcPool, errors, type, settings, batch, threads);
	th::thread readerThread(th::ref(dataReader));
	th::thread gpuThread(th::ref(worker));
	dataWriter();                                          // this thread writes, as in the reference
	readerThread.join(); gpuThread.join();
	reader.Close();
	writer.FinishCompress();
	if (errors.IsError()) { std::fprintf(stderr, "Error: %s\n", errors.GetError().c_str()); return 1; }
	return 0;
}

int decompress(const comp::InputParameters& args, uint32 batch)
{
	comp::DsrcFileReader reader;
	reader.StartDecompress(args.inputFilename);
	fq::FastqFileWriter writer(args.outputFilename);
	const uint32 threads = 1, parts = 512;
	comp::DsrcDataPool dsrcPool(parts, (uint64)args.fastqBufferSizeMB << 20);
	comp::DsrcDataQueue dsrcQueue(parts, 1);
	fq::FastqDataPool fastqPool(parts, comp::DsrcDataPool::DefaultBufferPartSize);
	fq::FastqDataQueue fastqQueue(parts, threads);
	core::ErrorHandler* eh = new core::ErrorHandler();
	core::ErrorHandler& errors = *eh;
	comp::DsrcReader dataReader(reader, dsrcQueue, dsrcPool, errors);
	fq::FastqWriter dataWriter(writer, fastqQueue, fastqPool, errors);
	DsrcDecompressorGpu worker(fastqQueue, fastqPool, dsrcQueue, dsrcPool, errors, reader.GetDatasetType(), reader.GetCompressionSettings(), batch, threads);
	th::thread readerThread(th::ref(dataReader));
	th::thread gpuThread(th::ref(worker));
	dataWriter();
	readerThread.join(); gpuThread.join();
	reader.FinishDecompress();
	writer.Close();
	if (errors.IsError()) { std::fprintf(stderr, "Error: %s\n", errors.GetError().c_str()); return 1; }
	return 0;
}

} // namespace

int main(int argc, char** argv)
{
	if (argc < 4 || (argv[1][0] != 'c' && argv[1][0] != 'd')) { std::fprintf(stderr, "usage: dsrc_ref_gpu <c|d> [options] in out\n"); return 2; }
	comp::InputParameters args;
	uint32 batch = 64;
	for (int i = 2; i < argc - 2; ++i)
	{
		const char* a = argv[i];
		if (a[0] != '-') continue;
		const int v = std::atoi(a + 2);
		switch (a[1])
		{
		case 'd': args.dnaCompressionLevel = v; break;
		case 'q': args.qualityCompressionLevel = v; break;
		case 'l': args.lossyCompression = true; break;
		case 'c': args.calculateCrc32 = true; break;
		case 'b': args.fastqBufferSizeMB = v; break;
		case 'o': args.qualityOffset = v; break;
		case 'n': batch = v > 0 ? (uint32)v : 1u; break;
		default: break;
		}
	}
	args.inputFilename = argv[argc - 2]; args.outputFilename = argv[argc - 1];
	try { return argv[1][0] == 'c' ? compress(args, batch) : decompress(args, batch); }
	catch (const std::exception& e) { std::fprintf(stderr, "Error: %s\n", e.what()); return 1; }
}
