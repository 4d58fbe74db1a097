/*
 * TEST INFRASTRUCTURE ONLY -- the parity oracle.
 *
 * A plain-C, single-threaded restatement of the DSRC 2 per-block compressor
 * (reference: /root/reference/src/BlockCompressor.cpp:208-259 and everything it
 * calls).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library; the product (dsrc_amd/) never does.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py checks this restatement
 * against the unmodified reference compiled into oracle/_ref/ (block bytes,
 * per-stream sizes, stats, primitives), and tests/golden/ holds vectors produced
 * by that reference build (tests/golden/make_golden.py).
 */
#ifndef DSRC_ORACLE_H
#define DSRC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_config
{
	uint32_t dna_order;          /* CompressionSettings::dnaOrder     (src/Common.h:118) = level*3 */
	uint32_t quality_order;      /* CompressionSettings::qualityOrder = level (lossless) / level*3 (lossy) */
	uint64_t tag_preserve_flags; /* -f mask: bit k set = keep title field k (1-based), 0 = keep titles as they are */
	int32_t  lossy;
	int32_t  calc_crc32;
	uint32_t quality_offset;     /* FastqDatasetType (src/Common.h:56-80) */
	int32_t  plus_repetition;
	int32_t  color_space;        /* SOLiD: primer base + colours (src/RecordsProcessor.cpp:25-58)   */
} orc_config;

enum { ORC_OK = 0, ORC_E_CAP = -1, ORC_E_UNSUPPORTED = -2, ORC_E_INPUT = -3, ORC_E_IO = -4 };

/* BlockCompressor::Store: one FASTQ chunk (no trailing newline) -> one block.
 * raw[4]/comp[4] follow fq::StreamsInfo order: Meta, Tag, Dna, Quality. */
int orc_compress_block(const orc_config* cfg, const uint8_t* in, uint64_t size,
					   uint8_t* out, uint64_t cap, uint64_t* out_size,
					   uint64_t raw[4], uint64_t comp[4]);

/* Same, for a BlockCompressor that has already coded blocks: *fields_cap carries the capacity
 * of TagStats::fields (std::vector<Field>, src/TagModeler.h:124) from block to block; 0 = fresh.
 * It decides which numeric fields keep record 0's double count in num_values (see tags_init). */
int orc_compress_block_state(const orc_config* cfg, uint32_t* fields_cap, const uint8_t* in, uint64_t size,
							 uint8_t* out, uint64_t cap, uint64_t* out_size,
							 uint64_t raw[4], uint64_t comp[4]);

/* ParseRecords + PreprocessRecords statistics (layout as ref_block_stats). */
int orc_block_stats(const orc_config* cfg, const uint8_t* in, uint64_t size,
					uint32_t* d_out, uint32_t* q_out, uint64_t* recs, uint64_t* chunk_size, uint64_t raw[4]);

/* FastqParser::Analyze (src/FastqParser.cpp:27-138). qoff in/out. */
int orc_analyze(const uint8_t* in, uint64_t size, int estimate_offset,
				uint32_t* qoff, int32_t* plus_rep, int32_t* color_space);

/* IFastqStreamReader::ReadNextChunk restated over an in-memory file
 * (src/FastqStream.cpp:18-98).  Fills starts[i]/sizes[i]; returns chunk count or <0. */
int64_t orc_cut_chunks(const uint8_t* file, uint64_t file_size, uint64_t buf_size,
					   uint64_t* starts, uint64_t* sizes, uint64_t cap);

/* DsrcFileWriter (src/DsrcFile.cpp:112-170): header + blocks + footer from block sizes. */
uint64_t orc_archive_header(uint8_t out[40], uint64_t footer_offset, uint32_t footer_size, uint64_t block_count);
uint64_t orc_archive_footer(uint8_t* out, const uint32_t* block_sizes, uint64_t block_count, const orc_config* cfg);

/* whole file: analyze + cut + compress every chunk + archive.  level semantics
 * as the CLI (-d, -q, -l, -c, -o, -b).  Single-threaded. */
int orc_compress_file(const char* in_path, const char* out_path, uint32_t dna_level, uint32_t quality_level,
					  int lossy, int crc, uint32_t qoff, uint32_t buf_mb);

/* wrap::DsrcArchive write path (record-level API) over a FASTQ file read like wrap::FastqFile does:
 * chunking by payload bytes, running chunkSize, settings mapping of src/DsrcArchive.cpp:33-47.
 * _block: one Flush() of BlockCompressorExt for a chunk given as text (no final newline) with that chunkSize word. */
int orc_compress_records_block(const orc_config* cfg, uint32_t* fields_cap, uint32_t chunk_size, const uint8_t* in, uint64_t size,
							   uint8_t* out, uint64_t cap, uint64_t* out_size, uint64_t raw[4], uint64_t comp[4]);
int orc_compress_records_file(const char* in_path, const char* out_path, uint32_t dna_level, uint32_t quality_level,
							  int lossy, uint32_t qoff, uint32_t buf_mb, int plus_rep);

/* BlockCompressor::Read (src/BlockCompressor.cpp:262-297): one block -> the FASTQ text of the chunk, every line
 * (also the last) ended by '\n'.  stored_crc / actual_crc (each tag, sequence, quality; may be NULL): the checksum
 * words of the block's meta stream and the ones recomputed over the decoded records as VerifyChecksum does
 * (src/BlockCompressor.cpp:576-594).  Implemented in dsrc_oracle_dec.c. */
int orc_decompress_block(const orc_config* cfg, const uint8_t* in, uint64_t size, uint8_t* out, uint64_t cap,
						 uint64_t* out_size, uint32_t stored_crc[3], uint32_t actual_crc[3]);
/* VerifyChecksum: 1 = the enabled checksums match, 0 = mismatch, < 0 = ORC_E_* */
int orc_verify_block(const orc_config* cfg, const uint8_t* in, uint64_t size, uint64_t text_cap);
uint32_t orc_crc32_update(uint32_t state, const uint8_t* p, uint32_t n);

/* primitives (same calling convention as the ref_* probes) */
uint64_t orc_bitwriter_script(const uint32_t* ops, uint32_t nops, uint8_t* out, uint64_t cap);
uint64_t orc_huffman(const uint32_t* freqs, uint32_t n, uint32_t* codes, uint32_t* lens, uint8_t* tree, uint64_t cap);
uint64_t orc_rc_script(const uint32_t* fct, uint32_t n, uint8_t* out, uint64_t cap);
uint64_t orc_rc_adaptive4(const uint8_t* syms, uint32_t n, uint8_t* out, uint64_t cap);
uint32_t orc_crc32(const uint8_t* p, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif
