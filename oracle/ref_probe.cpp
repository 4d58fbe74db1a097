// TEST INFRASTRUCTURE ONLY -- never linked into the product.
//
// Thin extern "C" driver over the UNMODIFIED reference sources under
// /root/reference/src (compiled where they lie by oracle/Makefile into
// oracle/_ref/libdsrc_ref.so).  It exposes the reference's own
// BlockCompressor::Store (src/BlockCompressor.cpp:208-220), the first-chunk
// FastqParser::Analyze (src/FastqParser.cpp:27-138), whole-file
// DsrcCompressorMT/DsrcDecompressorMT (src/DsrcOperator.cpp:230-521) and a few
// primitives (bit writer, Huffman, range coder) so that tests can pin
// oracle/dsrc_oracle.c and the HIP path against the real thing.
//
// This file contains no reference code: it only *calls* it.

#include <cstring>
#include <string>
#include <vector>

#include "BlockCompressor.h"
#include "BitMemory.h"
#include "FastqParser.h"
#include "FastqStream.h"
#include "DsrcOperator.h"
#include "huffman.h"
#include "RangeCoder.h"
#include "SymbolCoderRC.h"
#include "DnaModelerRCO.h"
#include "QualityOrderModeler.h"
#include "Crc32.h"

using namespace dsrc;

namespace {

struct ProbeCompressor : public comp::BlockCompressor
{
	ProbeCompressor(const fq::FastqDatasetType& t, const comp::CompressionSettings& s)
		: comp::BlockCompressor(t, s) {}

	// run only Parse + Preprocess and expose the stats (row a-3/a-4/a-5 fixtures)
	void StatsOnly(const fq::FastqDataChunk& chunk, fq::StreamsInfo& raw,
				   comp::DnaStats& d, comp::QualityStats& q, uint64& recs, uint64& chunkSize)
	{
		ParseRecords(chunk, raw);
		PreprocessRecords(chunkHeader.checksumFlags);
		d = recordsProcessor->GetDnaStats();
		q = recordsProcessor->GetQualityStats();
		recs = chunkHeader.recordsCount;
		chunkSize = chunkHeader.chunkSize;
		Reset();
	}
};

static comp::CompressionSettings mk_settings(uint32 dnaOrder, uint32 quaOrder, int lossy, int crc, uint64 tagFlags)
{
	comp::CompressionSettings s;
	s.dnaOrder = dnaOrder;
	s.qualityOrder = quaOrder;
	s.lossy = lossy != 0;
	s.calculateCrc32 = crc != 0;
	s.tagPreserveFlags = tagFlags;
	return s;
}

static fq::FastqDatasetType mk_dataset(uint32 qoff, int plusRep, int cs)
{
	fq::FastqDatasetType t;
	t.qualityOffset = qoff;
	t.plusRepetition = plusRep != 0;
	t.colorSpace = cs != 0;
	return t;
}

} // namespace

extern "C" {

// One chunk in -> one block out, exactly what DsrcCompressor::Process does per
// part (src/DsrcWorker.cpp:44-51).  Returns 0, or -1 if cap is too small.
int ref_compress_block(uint32 dnaOrder, uint32 quaOrder, int lossy, int crc, uint64 tagFlags,
					   uint32 qoff, int plusRep, int cs,
					   const uint8* in, uint64 size,
					   uint8* out, uint64 cap, uint64* outSize,
					   uint64* rawSizes, uint64* compSizes)
{
	fq::FastqDataChunk chunk(size + 64);
	std::memcpy(chunk.data.Pointer(), in, size);
	std::memset(chunk.data.Pointer() + size, '\n', 8);   // what follows the last title/quality in the real buffer is unspecified; tests never depend on it
	chunk.size = size;

	ProbeCompressor bc(mk_dataset(qoff, plusRep, cs), mk_settings(dnaOrder, quaOrder, lossy, crc, tagFlags));
	core::Buffer buf(size + (1 << 16));
	core::BitMemoryWriter w(buf);
	fq::StreamsInfo raw, comp;
	bc.Store(w, raw, comp, chunk);
	w.Flush();
	uint64 n = w.Position();
	*outSize = n;
	for (int i = 0; i < 4; ++i) { rawSizes[i] = raw.sizes[i]; compSizes[i] = comp.sizes[i]; }
	if (n > cap)
		return -1;
	std::memcpy(out, w.Pointer(), n);
	return 0;
}

// Block decode (src/BlockCompressor.cpp:262-297) -> FASTQ text of the chunk
// (with the trailing '\n' the reader re-adds).
int ref_decompress_block(uint32 dnaOrder, uint32 quaOrder, int lossy, int crc, uint64 tagFlags,
						 uint32 qoff, int plusRep, int cs,
						 const uint8* in, uint64 size, uint8* out, uint64 cap, uint64* outSize)
{
	comp::BlockCompressor bc(mk_dataset(qoff, plusRep, cs), mk_settings(dnaOrder, quaOrder, lossy, crc, tagFlags));
	std::vector<uint8> tmp(in, in + size);
	core::BitMemoryReader r(tmp.data(), size);
	fq::FastqDataChunk chunk(cap + 64);
	bc.Read(r, chunk);
	*outSize = chunk.size;
	if (chunk.size > cap)
		return -1;
	std::memcpy(out, chunk.data.Pointer(), chunk.size);
	return 0;
}

// stats after ParseRecords + PreprocessRecords.
// q_out: [symbolCount, minLength, maxLength, rawLength, thLength, rleLength] then 256 freqs
// d_out: [symbolCount] then 20 freqs
int ref_block_stats(int lossy, uint32 qoff, const uint8* in, uint64 size,
					uint32* d_out, uint32* q_out, uint64* recs, uint64* chunkSize, uint64* rawSizes)
{
	fq::FastqDataChunk chunk(size + 64);
	std::memcpy(chunk.data.Pointer(), in, size);
	std::memset(chunk.data.Pointer() + size, '\n', 8);
	chunk.size = size;
	ProbeCompressor bc(mk_dataset(qoff, 0, 0), mk_settings(0, 0, lossy, 0, 0));
	comp::DnaStats d; comp::QualityStats q; fq::StreamsInfo raw;
	bc.StatsOnly(chunk, raw, d, q, *recs, *chunkSize);
	d_out[0] = d.symbolCount;
	for (int i = 0; i < 20; ++i) d_out[1 + i] = d.symbolFreqs[i];
	q_out[0] = q.symbolCount; q_out[1] = q.minLength; q_out[2] = q.maxLength;
	q_out[3] = q.rawLength; q_out[4] = q.thLength; q_out[5] = q.rleLength;
	for (int i = 0; i < 256; ++i) q_out[6 + i] = q.symbolFreqs[i];
	for (int i = 0; i < 4; ++i) rawSizes[i] = raw.sizes[i];
	return 0;
}

// FastqParser::Analyze on the first chunk.
int ref_analyze(const uint8* in, uint64 size, int estimateOffset, uint32* qoff, int* plusRep, int* cs)
{
	fq::FastqDataChunk chunk(size + 64);
	std::memcpy(chunk.data.Pointer(), in, size);
	chunk.size = size;
	fq::FastqDatasetType t;
	t.qualityOffset = *qoff;
	fq::FastqParser p;
	bool ok = p.Analyze(chunk, t, estimateOffset != 0);
	*qoff = t.qualityOffset; *plusRep = t.plusRepetition; *cs = t.colorSpace;
	return ok ? 0 : -1;
}

// Whole file through the reference pipeline (1 reader + T workers + writer).
int ref_compress_file(const char* in, const char* out, uint32 dnaLevel, uint32 quaLevel, int lossy, int crc,
					  uint32 qoff, uint32 bufMB, uint32 threads, uint64 tagFlags)
{
	comp::InputParameters p;
	p.inputFilename = in; p.outputFilename = out;
	p.dnaCompressionLevel = dnaLevel; p.qualityCompressionLevel = quaLevel;
	p.lossyCompression = lossy != 0; p.calculateCrc32 = crc != 0;
	p.qualityOffset = qoff; p.fastqBufferSizeMB = bufMB; p.threadNum = threads;
	p.tagPreserveFlags = tagFlags;
	comp::IDsrcOperator* op = (threads <= 1) ? (comp::IDsrcOperator*)new comp::DsrcCompressorST()
											  : (comp::IDsrcOperator*)new comp::DsrcCompressorMT();
	bool ok = op->Process(p);
	delete op;
	return ok ? 0 : -1;
}

int ref_decompress_file(const char* in, const char* out, uint32 threads)
{
	comp::InputParameters p;
	p.inputFilename = in; p.outputFilename = out; p.threadNum = threads;
	comp::IDsrcOperator* op = (threads <= 1) ? (comp::IDsrcOperator*)new comp::DsrcDecompressorST()
											  : (comp::IDsrcOperator*)new comp::DsrcDecompressorMT();
	bool ok = op->Process(p);
	delete op;
	return ok ? 0 : -1;
}

// Chunk cutter (src/FastqStream.cpp:18-72): returns the sizes of all chunks of a file.
int ref_chunk_sizes(const char* in, uint32 bufMB, uint64* sizes, uint32 cap, uint32* count)
{
	fq::FastqFileReader rd(in);
	fq::FastqDataChunk chunk((uint64)bufMB << 20);
	uint32 n = 0;
	while (rd.ReadNextChunk(&chunk))
	{
		if (n < cap) sizes[n] = chunk.size;
		n++;
	}
	rd.Close();
	*count = n;
	return 0;
}

// ---- primitives ---------------------------------------------------------

// script: ops[i] = {kind, a, b}: 0 PutBit(a) 1 Put2Bits(a) 2 PutBits(a,b) 3 PutByte(a) 4 PutWord(a) 5 FlushPartial
uint64 ref_bitwriter_script(const uint32* ops, uint32 nops, uint8* out, uint64 cap)
{
	core::BitMemoryWriter w((uint32)(cap > 64 ? cap : 64));
	for (uint32 i = 0; i < nops; ++i)
	{
		uint32 k = ops[3*i], a = ops[3*i+1], b = ops[3*i+2];
		switch (k)
		{
			case 0: w.PutBit(a); break;
			case 1: w.Put2Bits(a); break;
			case 2: w.PutBits(a, b); break;
			case 3: w.PutByte((byte)a); break;
			case 4: w.PutWord(a); break;
			case 5: w.FlushPartialWordBuffer(); break;
		}
	}
	w.Flush();
	uint64 n = w.Position();
	std::memcpy(out, w.Pointer(), n < cap ? n : cap);
	return n;
}

// Restart(n); Insert(freq[i]); Complete(); StoreTree -> tree bytes; codes/lens for ids < n
uint64 ref_huffman(const uint32* freqs, uint32 n, uint32* codes, uint32* lens, uint8* tree, uint64 cap)
{
	comp::HuffmanEncoder h;
	h.Restart(n);
	for (uint32 i = 0; i < n; ++i) h.Insert(freqs[i]);
	h.Complete();
	const comp::HuffmanEncoder::Code* c = h.GetCodes();
	for (uint32 i = 0; i < n; ++i) { codes[i] = c[i].code; lens[i] = c[i].len; }
	core::BitMemoryWriter w((uint32)(cap > 64 ? cap : 64));
	h.StoreTree(w);
	w.Flush();
	uint64 sz = w.Position();
	std::memcpy(tree, w.Pointer(), sz < cap ? sz : cap);
	return sz;
}

// raw (freq, cum, total) script through RangeEncoder
uint64 ref_rc_script(const uint32* fct, uint32 n, uint8* out, uint64 cap)
{
	core::BitMemoryWriter w((uint32)(cap > 64 ? cap : 64));
	comp::RangeEncoder rc(w);
	rc.Start();
	for (uint32 i = 0; i < n; ++i) rc.EncodeFrequency(fct[3*i], fct[3*i+1], fct[3*i+2]);
	rc.End();
	w.Flush();
	uint64 sz = w.Position();
	std::memcpy(out, w.Pointer(), sz < cap ? sz : cap);
	return sz;
}

// one TSymbolCoderRC<4> over a symbol sequence (rescale KAT)
uint64 ref_rc_adaptive4(const uint8* syms, uint32 n, uint8* out, uint64 cap)
{
	core::BitMemoryWriter w((uint32)(cap > 64 ? cap : 64));
	comp::RangeEncoder rc(w);
	comp::TSymbolCoderRC<4> coder;
	rc.Start();
	for (uint32 i = 0; i < n; ++i) coder.EncodeSymbol(rc, syms[i]);
	rc.End();
	w.Flush();
	uint64 sz = w.Position();
	std::memcpy(out, w.Pointer(), sz < cap ? sz : cap);
	return sz;
}

uint32 ref_crc32(const uint8* p, uint32 n)
{
	core::Crc32Hasher h;
	return h.ComputeHash(p, n);
}

} // extern "C"
