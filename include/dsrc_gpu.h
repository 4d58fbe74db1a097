/*
 * dsrc_gpu.h -- C ABI of the MI355X-native DSRC block-compression path (libdsrc_gpu.so).
 *
 * This is the drop-in boundary: plain C, opaque handle, pointers and sizes only.  It replaces the
 * reference's CPU worker pool -- N threads each running
 *     BlockCompressor(datasetType, compSettings)                  (src/BlockCompressor.h:66)
 *     BlockCompressor::Store(BitMemoryWriter&, StreamsInfo& raw,
 *                            StreamsInfo& comp, const FastqDataChunk&)   (src/BlockCompressor.h:69)
 * inside DsrcCompressor::Process (src/DsrcWorker.cpp:30-73) -- by one GPU block scheduler.  Every
 * function returns 0 on success or a negative DSRCGPU_E_* code; dsrcgpu_last_error() gives the text.
 * Output blocks are bit-identical to what BlockCompressor::Store writes for the same chunk.
 *
 * INTEGRATION.md shows the binding a reference maintainer would add on top of this header.
 */
#ifndef DSRC_GPU_H
#define DSRC_GPU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dsrcgpu_handle dsrcgpu_handle;

/* comp::CompressionSettings (src/Common.h:110-147).  Orders, not CLI levels:
 * dna_order = 3*level; quality_order = level (lossless) or 3*level (lossy) -- IDsrcOperator::GetCompressionSettings
 * (src/DsrcOperator.h:74-90). */
typedef struct dsrcgpu_settings
{
	uint32_t dna_order;
	uint32_t quality_order;
	uint64_t tag_preserve_flags;   /* -f mask (FastqParserExt, src/FastqParser.cpp:167-251): bit k set = keep title field k, k = 1..30;
	                                * 0 = titles as they are.  With a mask the chunk text is rewritten in place (device copy / the
	                                * caller's device buffer in the *_device entry point), as BlockCompressor::Store does to its input */
	uint8_t  lossy;
	uint8_t  calculate_crc32;
	uint8_t  verify_after_compress;/* with calculate_crc32: every compress call decodes the blocks it has written (on the device, nothing is
	                                * copied back) and fails with DSRCGPU_E_CRC "CRC32 checksums mismatch." unless the stored checksums match
	                                * -- the reference's worker does this whenever -c is given (src/DsrcWorker.cpp:53-62) */
	uint8_t  reserved[5];
} dsrcgpu_settings;

/* fq::FastqDatasetType (src/Common.h:56-80), decided once per file by FastqParser::Analyze on chunk 0. */
typedef struct dsrcgpu_dataset
{
	uint32_t quality_offset;       /* 33..64, already resolved (not 0/auto) */
	uint8_t  plus_repetition;
	uint8_t  color_space;          /* SOLiD colour space (primer base + colours); not together with tag_preserve_flags / record layout */
	uint8_t  reserved[2];
} dsrcgpu_dataset;

enum
{
	DSRCGPU_OK            =  0,
	DSRCGPU_E_ARG         = -1,   /* bad argument / unsupported setting */
	DSRCGPU_E_HIP         = -2,   /* HIP runtime failure (text in last_error) */
	DSRCGPU_E_NOMEM       = -3,
	DSRCGPU_E_CAPACITY    = -4,   /* caller's output buffer too small */
	DSRCGPU_E_INPUT       = -5,   /* a chunk could not be coded (no records, invalid bases, reference-UB input ...) */
	DSRCGPU_E_STATE       = -6,
	DSRCGPU_E_CRC         = -7,   /* verify_after_compress: a block did not decode back to the checksums stored in it */
	DSRCGPU_E_BUSY        = -8    /* dsrcgpu_submit: all batches of the ring are in flight -- collect and release blocks, then submit again */
};

/* Replaces: BlockCompressor::BlockCompressor (src/BlockCompressor.cpp:53-94) x worker threads.
 * device: HIP device ordinal.  arena_bytes: HBM to reserve for batch scratch, 0 = grow on demand. */
int dsrcgpu_create(const dsrcgpu_settings* settings, const dsrcgpu_dataset* dataset, int device,
				   uint64_t arena_bytes, dsrcgpu_handle** out);
void dsrcgpu_destroy(dsrcgpu_handle* h);
const char* dsrcgpu_last_error(const dsrcgpu_handle* h);

/* Replaces one call of BlockCompressor::Store (src/BlockCompressor.cpp:208-220) + bitMemory.Flush()
 * (src/DsrcWorker.cpp:48-51): chunk (host memory, size = FastqDataChunk::size, i.e. without the final newline)
 * -> block bytes.  raw_sizes/comp_sizes are fq::StreamsInfo::sizes in enum order Meta, Tag, Dna, Quality
 * (src/Common.h:82-105).  Synchronous. */
int dsrcgpu_compress_block(dsrcgpu_handle* h, const uint8_t* fastq, uint64_t size,
						   uint8_t* block, uint64_t block_cap, uint64_t* block_size,
						   uint64_t raw_sizes[4], uint64_t comp_sizes[4]);

/* The same for n chunks in one scheduler pass (this is what keeps the GPU full).  Blocks are written back to
 * back into `blocks` in chunk order; block_offs/block_sizes get n entries, raw_sizes/comp_sizes 4*n.
 * Compressor state that the reference carries from block to block inside one BlockCompressor (the capacity of
 * TagStats::fields) advances in chunk order, as with `dsrc c -t1`. */
int dsrcgpu_compress_batch(dsrcgpu_handle* h, uint32_t n, const uint8_t* const* fastq, const uint64_t* sizes,
						   uint8_t* blocks, uint64_t blocks_cap, uint64_t* block_offs, uint64_t* block_sizes,
						   uint64_t* raw_sizes, uint64_t* comp_sizes);

/* Device-resident variant: d_fastq is a HIP device pointer; chunk i is [offs[i], offs[i]+sizes[i]).
 * d_blocks (device, blocks_cap bytes) receives the blocks back to back.  No host<->device
 * payload copies happen inside this call; it is what bench.py times. */
int dsrcgpu_compress_batch_device(dsrcgpu_handle* h, uint32_t n, const void* d_fastq, const uint64_t* offs,
								  const uint64_t* sizes, void* d_blocks, uint64_t blocks_cap,
								  uint64_t* block_offs, uint64_t* block_sizes,
								  uint64_t* raw_sizes, uint64_t* comp_sizes);

/* Scheduler lanes inside a handle (round 6).  The reference needs one DsrcCompressorMT for a file (src/DsrcOperator.cpp:295-340);
 * a batch call on ONE handle now fills the GPU by itself: a batch of more than one sub-batch's worth of chunks is cut into
 * consecutive sub-batches (default: about 1.9 GB of chunks each) that run on up to `lanes` (default 4) scheduler lanes inside the
 * handle -- own arena and streams each, one host thread per lane for the duration of the call -- so that the serial range coder
 * of one sub-batch runs beside the front ends of the others.  The block-to-block state goes from sub-batch to sub-batch through an
 * internal chain: the blocks, their order and the state left behind are those of the same call on one lane.  HBM: an arena per
 * lane, sized for a sub-batch (about 7.5 x its chunks + 1.75 GiB), instead of one sized for the batch.
 * dsrcgpu_set_lanes(h, lanes, sub_batch_chunks): 0 = the default for either; lanes = 1 keeps a batch on the handle's own lane.
 * A handle that has been given a chain (dsrcgpu_set_chain) or a fixed arena (dsrcgpu_create's arena_bytes) is the caller's own
 * lane and is never cut; nor are the batches of the queue form's lanes. */
int dsrcgpu_set_lanes(dsrcgpu_handle* h, uint32_t lanes, uint32_t sub_batch_chunks);

/* ---- decompression -------------------------------------------------------------------------------------------------
 * Replaces one call of BlockCompressor::Read (src/BlockCompressor.cpp:262-297) inside DsrcDecompressor::Process
 * (src/DsrcWorker.cpp:75-104): block bytes -> the FASTQ text of the chunk, every line (also the last) ended by '\n'.
 * The handle's settings and dataset must be the archive's (DsrcFileFooter, src/DsrcFile.cpp:142-170); a block that does
 * not end exactly where its last stream ends is refused (DSRCGPU_E_INPUT), as are blocks the reference's own decoder
 * cannot read back deterministically (colour space without a constant primer; streams that run off the end of the block).
 * crc_ok (may be NULL): per block, 1 if the checksum words stored with calculate_crc32 equal the checksums of the
 * decoded records, 0 otherwise -- BlockCompressor::VerifyChecksum (src/BlockCompressor.cpp:576-594), which the
 * reference's compressing worker runs on every block it has just written (src/DsrcWorker.cpp:53-62); without
 * calculate_crc32 every entry is 1.
 * text_caps (may be NULL): text bytes to reserve per block instead of the chunkSize word + 1; needed for archives written
 * by the record-level API, whose chunkSize words are running totals (see dsrcgpu_set_record_layout).
 * n blocks in one scheduler pass; texts are laid out back to back in block order (text_offs / text_sizes). */
int dsrcgpu_decompress_block(dsrcgpu_handle* h, const uint8_t* block, uint64_t size,
							 uint8_t* text, uint64_t text_cap, uint64_t* text_size, uint32_t* crc_ok);
int dsrcgpu_decompress_batch(dsrcgpu_handle* h, uint32_t n, const uint8_t* const* blocks, const uint64_t* sizes,
							 const uint64_t* text_caps, uint8_t* text, uint64_t text_cap,
							 uint64_t* text_offs, uint64_t* text_sizes, uint32_t* crc_ok);
/* Device-resident variant (block i is d_blocks[offs[i] .. offs[i] + sizes[i])); no payload copies inside the call. */
int dsrcgpu_decompress_batch_device(dsrcgpu_handle* h, uint32_t n, const void* d_blocks, const uint64_t* offs,
									const uint64_t* sizes, const uint64_t* text_caps, void* d_text, uint64_t text_cap,
									uint64_t* text_offs, uint64_t* text_sizes, uint32_t* crc_ok);

/* Queue form of DsrcCompressor::Process (src/DsrcWorker.cpp:39-70):
 *   fastqQueue.Pop(partId, chunk)            -> dsrcgpu_submit(partId, chunk)      (bytes are copied into page-locked staging)
 *   ... Store ... dsrcQueue.Push(partId, blk) -> dsrcgpu_collect(&partId, &blk, ...)
 *   dsrcPool.Release(blk)                    -> dsrcgpu_release(blk)
 * Asynchronous: dsrcgpu_flush hands everything submitted since the last flush to the handle's scheduler lanes as one
 * batch and returns; the ring holds as many batches as the handle has scheduler lanes, plus two: one being filled, the others running,
 * waiting or being collected.  When all are busy
 * dsrcgpu_submit returns DSRCGPU_E_BUSY without copying anything (it does not wait: the caller may be the thread that has
 * to collect): take blocks with dsrcgpu_collect, release them, submit again -- a slot is free once every block of the
 * oldest batch has been released.  Blocks come back in submission order:
 * dsrcgpu_collect returns 1 and a block (a pointer into page-locked memory owned by the handle, valid until
 * dsrcgpu_release), waits while a flushed batch is still running, and returns 0 when everything flushed has been
 * collected; dsrcgpu_try_collect never waits (0 = nothing ready right now).  A batch that failed makes the next call
 * return its error.  One submitter thread and one collector thread may use a handle concurrently; the batch calls above
 * must not be mixed in while batches are in flight.  Block-to-block state follows submission order (`dsrc c -t1`).
 * The batches that run at a time run on scheduler lanes inside the handle (round 4: two, round 6: DSRC_GPU_QUEUE_LANES, default 3,
 * at most 4) -- the handle and twins with their own arenas and streams, created at the first flush -- so that the range coder of one
 * batch (~0.08 s on a few CUs, whatever the batch's size) overlaps the copies and the front ends of the next ones; the block-to-block
 * state goes from lane to lane in flush order through an internal chain.  Every lane adds an arena (about 7.5 x a batch's chunks + 1.75
 * GiB) to the handle's HBM; the twins are left out when the caller has given the handle a chain of his own (dsrcgpu_set_chain) or with
 * DSRC_GPU_QUEUE_LANES=1.  dsrcgpu_set_fields_capacity counts before the first flush and between flushes once the queue has drained
 * (DSRCGPU_E_STATE while batches are in flight); dsrcgpu_set_record_layout belongs to whichever comes first, the next flush or the
 * next batch call, and travels with that batch: it may be set for the next flush while earlier batches are still running. */
int dsrcgpu_submit(dsrcgpu_handle* h, int64_t part_id, const uint8_t* fastq, uint64_t size);
/* The same without the copy (round 6): `fastq` is page-locked memory of the caller's (dsrcgpu_host_alloc) and stays as it is until
 * every block of its batch has been collected -- the batch's copy to the device reads it in place.  A submitter that copies 8 MiB chunks
 * into the ring moves ~10 GB/s; the reader threads of a host pipeline can fill page-locked buffers directly instead.  Both forms
 * may be mixed in one batch. */
int dsrcgpu_submit_pinned(dsrcgpu_handle* h, int64_t part_id, const uint8_t* fastq, uint64_t size);
int dsrcgpu_flush(dsrcgpu_handle* h);
int dsrcgpu_collect(dsrcgpu_handle* h, int64_t* part_id, uint8_t** block, uint64_t* block_size,
					uint64_t raw_sizes[4], uint64_t comp_sizes[4]);
int dsrcgpu_try_collect(dsrcgpu_handle* h, int64_t* part_id, uint8_t** block, uint64_t* block_size,
						uint64_t raw_sizes[4], uint64_t comp_sizes[4]);
int dsrcgpu_release(dsrcgpu_handle* h, uint8_t* block);

/* Several handles may compress consecutive batches of ONE archive at the same time (one host thread each); this is
 * what hides the serial range-coder stage.  The only state DSRC carries from block to block -- the capacity of
 * TagStats::fields inside one BlockCompressor (src/TagModeler.h:124) -- is then handed from
 * batch `seq` to batch `seq + 1` through a chain, so the blocks equal those of a single handle fed in order, i.e.
 * `dsrc c -t1`.  dsrcgpu_set_chain(h, chain, seq) declares that the NEXT batch call on h is batch number `seq`
 * (0, 1, 2, ... without gaps across all handles of the chain); that call waits, early in its course, for batch
 * seq - 1 to have published the state.  A batch that fails marks the chain failed and releases the waiters. */
typedef struct dsrcgpu_chain dsrcgpu_chain;
int dsrcgpu_chain_create(dsrcgpu_chain** out);
void dsrcgpu_chain_destroy(dsrcgpu_chain* c);
int dsrcgpu_set_chain(dsrcgpu_handle* h, dsrcgpu_chain* c, uint64_t seq);

/* Sharding one archive over several devices or processes.  Chunk ranges are independent except for the one
 * value above, which after a chunk is a pure function of the value before it and the number of fields in the title of
 * the chunk's first record:  cap' = dsrcgpu_fields_capacity_after(cap, dsrcgpu_title_fields(title, len, flags)).
 * So a shard that starts at chunk k needs only the fold of that function over the first titles of chunks 0..k-1 -- a
 * host-side pass over one line per chunk (or an exclusive scan of one uint32 across ranks, dsrc_amd/dist.py) -- to write
 * exactly the blocks `dsrc c -t1` writes: seed the first handle (or the chain) of the shard with it. */
uint32_t dsrcgpu_title_fields(const uint8_t* title, uint32_t len, uint64_t tag_preserve_flags);
uint32_t dsrcgpu_fields_capacity_after(uint32_t cap, uint32_t n_fields);
int dsrcgpu_set_fields_capacity(dsrcgpu_handle* h, uint32_t cap);
int dsrcgpu_get_fields_capacity(const dsrcgpu_handle* h, uint32_t* cap);
int dsrcgpu_chain_seed(dsrcgpu_chain* c, uint32_t fields_capacity);      /* before batch 0 of the chain has run */

/* Record-level API (reference: wrap::BlockCompressorExt::WriteNextRecord/Flush, src/BlockCompressorExt.cpp:20-46,65-127,
 * used by wrap::DsrcArchive, src/DsrcArchive.cpp:129-150,217-224).  The caller assembles each chunk as FASTQ text
 * (tag\nsequence\nplus\nquality, no newline after the last record) and declares, for the NEXT batch call on h
 * (n must equal that call's chunk count; one-shot), that the chunks come from records:
 *   - block i stores chunk_sizes[i] as chunkSize (the reference keeps a running total over the whole archive there,
 *     because BlockCompressor::Reset does not clear it);
 *   - the reference lays tag, sequence and quality out back to back, so its tag tokenizer takes the first sequence
 *     byte (already turned into a base index) as the separator after the last title field; reproduced.
 * Not combinable with tag_preserve_flags or calculate_crc32 (the reference's archive API drops both). */
int dsrcgpu_set_record_layout(dsrcgpu_handle* h, uint32_t n, const uint32_t* chunk_sizes);

/* Decoding the order-context levels keeps one adaptive model table per block in flight (up to 64 MiB at -q2, DESIGN.md
 * section 11); by default a pass takes 70 % of the HBM that is free when it starts.  Hosts that run several decoding
 * handles on one device give each its share: dsrcgpu_set_table_budget(h, bytes) (0 = automatic again). */
int dsrcgpu_set_table_budget(dsrcgpu_handle* h, uint64_t bytes);
int dsrcgpu_device_memory(int device, uint64_t* free_bytes, uint64_t* total_bytes);
/* A handle keeps its batch arena and its table region between calls (tens of GB after a large pass).  A host that moves on to
 * another phase on the same device (other handles, other batch sizes) hands them back with dsrcgpu_release_memory; the next call
 * allocates again.  Not to be called while a call on this handle is in flight (queue form: after the last collect).  No
 * counterpart in the reference: its workers' buffers live as long as the workers. */
int dsrcgpu_release_memory(dsrcgpu_handle* h);
/* The opposite: a host that knows what its next call will need has the batch arena and / or the decoder's table region grown to at least
 * these sizes NOW (nothing shrinks; the table figure is capped by the table budget) -- on a side thread, or while it is still reading its
 * input.  Worth it where allocation is not free: HBM that another process has just released is wiped by the driver before it is handed
 * out again (~20-35 GB/s on the measured box, NOTES/round_5.md), and the first call would otherwise wait for that in the middle of its
 * course.  Not to be called while a call on this handle is in flight. */
int dsrcgpu_reserve_memory(dsrcgpu_handle* h, uint64_t arena_bytes, uint64_t table_bytes);

/* Optional: brings up the HIP runtime and the device context (the first HIP call of a process costs 0.3-1 s); call it on a
 * side thread while the host opens its files.  It is also the opt-in for the queue setting several handles per device need:
 * when the process has not set GPU_MAX_HW_QUEUES, the first call sets it to 24 -- effective only if this is the process's first
 * HIP call (the runtime reads the variable when it starts).  The library never touches the environment otherwise. */
int dsrcgpu_prepare(int device);

/* Page-locked host memory for chunk / block buffers: host<->device copies from it run at PCIe speed and
 * asynchronously to the other handles' kernels (the entry points accept any host pointer; pageable ones are slower). */
int dsrcgpu_host_alloc(uint64_t bytes, void** out);
int dsrcgpu_host_free(void* p);

/* Device arithmetic self-test: the exact-division identities the range-coder stage relies on (reciprocal of every
 * possible model total, quotients on a spread of numerators) are checked against the hardware integer division.
 * *mismatches must come back 0. */
int dsrcgpu_selftest(dsrcgpu_handle* h, uint32_t* mismatches);

/* Timing of the last batch measured with HIP events on the scheduler's stream: total ms of the batch's
 * kernels, ms of the range-coder kernel (k_rc), number of k_rc launches. */
int dsrcgpu_last_timing(const dsrcgpu_handle* h, float* batch_ms, float* rc_ms, uint32_t* rc_launches);

/* ... and of the two data-parallel stages that bound the throughput of the order-context levels: summed HIP-event time
 * of the k_sort launches (context sort) and of the k_replay_seams + k_replay launches (model replay) of the last batch. */
int dsrcgpu_last_stage_timing(const dsrcgpu_handle* h, float* sort_ms, float* replay_ms);

/* Counter-based synthetic Illumina-like FASTQ generated directly in HBM (bench input; same bytes as
 * dsrc_amd/synth.py illumina_fastq).  Writes records first..first+count-1, returns the byte count. */
int dsrcgpu_synth_illumina(dsrcgpu_handle* h, uint64_t first, uint64_t count, void* d_out, uint64_t cap, uint64_t* bytes);
/* ... with a flavour: 0 = the generator above (BASELINE's configurations); 1 = the same records with the qualities quantised to four
 * levels (Phred 2 / 12 / 23 / 37: what current instruments write) -- dsrc_amd/synth.py illumina_fastq(binned=True); bench.py's second line. */
int dsrcgpu_synth_fastq(dsrcgpu_handle* h, uint32_t flavour, uint64_t first, uint64_t count, void* d_out, uint64_t cap, uint64_t* bytes);

/* small HBM helpers so that non-HIP hosts (Python/ctypes, JNI ...) can stage device-resident batches */
int dsrcgpu_dev_alloc(dsrcgpu_handle* h, uint64_t bytes, void** d_ptr);
int dsrcgpu_dev_free(dsrcgpu_handle* h, void* d_ptr);
int dsrcgpu_dev_upload(dsrcgpu_handle* h, void* d_dst, const void* src, uint64_t bytes);
int dsrcgpu_dev_download(dsrcgpu_handle* h, void* dst, const void* d_src, uint64_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* DSRC_GPU_H */
