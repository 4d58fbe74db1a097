// Shared host/device data layout of the MI355X block-compression path.
//
// One *batch* = B independent FASTQ chunks ("blocks", reference
// fq::FastqDataChunk, src/Fastq.h:29) resident in HBM.  Everything a kernel
// needs about block b lives in BlkDesc[b] (host-written layout) and
// BlkState[b] (device-written results); bulk arrays are carved from one arena.
#pragma once
#include <stdint.h>

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef int64_t i64;

#ifndef DSRC_WG
#define DSRC_WG 1024            // threads per workgroup for the block-owning kernels (16 waves)
#endif
#define DSRC_WAVES (DSRC_WG / 64)
#define DSRC_LANE_BYTES 64               // bytes of FASTQ text one lane classifies (eight 8-byte words)
#define DSRC_TILE_BYTES (DSRC_WG * DSRC_LANE_BYTES)   // bytes one workgroup indexes

#define DSRC_MAX_FIELDS 64      // read-id fields per title (reference stores the count in one byte)
#define DSRC_MAX_STRF 129       // per-position statistics of a string field: 128 positions + overflow bucket

// error bits (BlkState::err); any bit => the block is reported as failed, never silently "fixed"
enum
{
	DSRC_ERR_NO_RECORDS   = 1u << 0,
	DSRC_ERR_BAD_BASE     = 1u << 1,   // sequence byte outside the reference's 19-symbol LUT (src/RecordsProcessor.cpp:186-206)
	DSRC_ERR_LONG_LINE    = 1u << 2,   // line > 65535 bytes (reference lengths are uint16, src/Fastq.h:37-40)
	DSRC_ERR_TOO_MANY_FLD = 1u << 3,
	DSRC_ERR_OUT_OVERFLOW = 1u << 4,
	DSRC_ERR_REF_UB       = 1u << 5,   // input drives the reference into undefined behaviour
	DSRC_ERR_CODE_TOO_LONG= 1u << 6,   // Huffman code > 31 bits (reference PutBits limit, src/BitMemory.h:318-338)
	DSRC_ERR_ARENA        = 1u << 7,
};

struct DsrcParams   // uniform over a batch
{
	u32 dna_order, quality_order;
	u32 lossy, crc;
	u32 quality_offset;
	u32 n_blocks;
	u32 max_tiles;          // tiles per block upper bound (grid.x of the tile kernels)
	u32 tag_flags;          // -f mask (bit k: keep title field k, 1-based); 0 = titles as they are
	u32 record_layout;      // chunks assembled by the record-level API (BlockCompressorExt): see dsrcgpu_set_record_layout
	u32 color_space;        // SOLiD: primer base + colours (src/RecordsProcessor.cpp:25-58)
};

// numeric-field coding schemes, Field::NumericSchemeEnum (src/TagModeler.h:73)
enum { NS_NONE = 0, NS_VALUE_VAR = 1, NS_VALUE_RLE = 2, NS_DELTA_VAR = 3, NS_DELTA_RLE = 4, NS_DELTA_CONST = 5 };

struct TagField
{
	// template from record 0 (TagAnalyzer::InitializeFieldsStats, src/TagModeler.cpp:159-222)
	u32 start0, len0;       // field text of record 0, relative to its title
	u8  sep;
	u8  isnum0;             // numeric in record 0
	u8  num_slot;           // index into the per-record value arrays (isnum0 fields only)
	u8  keep_double;        // record 0 counted twice in num_values (vector-capacity quirk, see dsrc_oracle.c tags_init)
	// reductions over all records (UpdateFieldsStats, :224-339)
	u32 min_len, max_len;
	u32 not_const, not_lenconst, not_numeric;
	i32 min_value, max_value, min_delta, max_delta;
	u32 runs_val, last_val_len;       // chunked value runs:  R chunks, length-1 of the last one
	u32 runs_delta, last_delta_len;
	// decisions (FinalizeFieldsStats, :461-551)
	u8  is_constant, is_numeric, is_len_constant, scheme;
	u8  var_stat_encode, is_string, pad0, pad1;
	u32 bits_num, bits_value, bits_len;
	// resources
	u32 hist_off;           // u32 index into the tag scratch: numeric histogram (<=512) or string [129][256]
	u32 code_off;           // u32 index: code tables (code,len pairs)
	u32 tree_off, tree_bytes; // serialized Huffman trees for the dictionary (bytes, in the tag scratch)
	u32 ham_off;            // byte offset: per-position hamming mask (string fields)
};

struct BlkDesc   // host -> device
{
	u64 in_off;  u32 in_size;  u32 n_tiles;
	u32 line_base;          // into the line-start pool (u32)
	u32 rec_base;           // into the per-record pools
	u32 rec_cap;
	u32 fields_keep_from;   // first field index whose record-0 double count survives
	u32 chunk_size_value;   // record layout: the chunkSize word of the block's meta stream (running total, src/BlockCompressorExt.cpp:126)
	u32 pad0;
	u64 q_base, d_base;     // quality / DNA symbol streams (bytes)
	// per-stream staging of the compressed block (u32 words, MSB-first "logical big-endian")
	u64 tag_out, qua_out, dna_out;        // u32 index
	u32 tag_cap, qua_cap, dna_cap;        // words
	u32 q_scheme;           // host-decided stream schemes (IQualityModelerProxy / IDnaModelerProxy::SelectSchemeId)
	u32 d_scheme;
	u32 plain_mask;         // bit0: quality stream staged as plain bytes, bit1: DNA stream (range-coder output)
	u64 out_off;            // final block position in the output buffer (bytes), set after sizes are known
};

struct BlkState  // device -> host (and device scratch)
{
	u32 err;
	u32 n_term, n_crlf, n_lines;
	i32 title_cut;          // -f: title bytes removed (FastqParserExt::totalBytesCut; -1 per record whose kept last field took its terminator along)
	u32 first_bad, n_recs;
	u32 q_total, d_total;
	u32 raw_tag, raw_dna, raw_qua;      // fq::StreamsInfo raw sizes (src/FastqParser.cpp:152-158)
	// QualityStats / DnaStats (src/Stats.h:44-101)
	u32 d_count, q_count;
	u32 d_freq[20];
	u32 q_freq[256];
	u8  d_sym[20];  u8 pad_[4];
	u8  q_sym[256];
	u32 min_len, max_len, raw_len, th_len, rle_len;
	u32 crc_tag, crc_seq, crc_qua;
	// tags
	u32 n_fields, n_num0;
	u32 mixed, first_mixed;
	u32 min_title, max_title;
	// schemes + sizes
	u32 flags;
	u32 meta_bytes, tag_bytes, qua_bytes, dna_bytes;
	u32 tag_hdr_bytes;
	u32 q_runs;             // RLE quality: number of runs
	u32 rle_qn;             // RLE quality: the modeler's own alphabet = values that start a run (QualityRLEModeler::qSymbols); differs
	u8  rle_qsym[256];      //   from q_sym only when records were shortened after the statistics (colour space)
	// colour space (ColorSpaceStats src/Stats.h:23-42; ChunkHeader::csSeqBegin/csQuaBegin src/BlockCompressor.h:43-44)
	u32 cs_varbegin;        // != 0: the records do not all start with record 0's primer character
	u32 cs_reduced;         // 1: constant primer, records were shortened by k_cs_reduce (FLAG_DELTA_CONSTANT)
	u32 cs_seq_begin, cs_qua_begin;
	u32 scratch[8];
	TagField fld[DSRC_MAX_FIELDS];
};
