/* jtokkit_amd.h -- C ABI of the MI355X-native batch BPE encoder.
 *
 * This is the drop-in boundary for ONE path of JTokkit (reference at /root/reference, paths below
 * relative to lib/src/main/java/com/knuddels/jtokkit/): GptBytePairEncoding.encode() and its
 * callers in the Encoding interface.  The reference has no FFI of its own; these are the entry
 * points a JNI shim behind `com.knuddels.jtokkit.api.Encoding` binds (see INTEGRATION.md for the
 * Java/JNI side).  Plain C: pointers and sizes only, no C++/torch types.
 *
 * Threading (api/EncodingRegistry.java:51,61 "The encoding must be thread-safe"): a jtk_encoding
 * is immutable after creation and may be shared by any number of threads.  A jtk_batch owns one
 * HIP stream plus its device scratch and must be used by one thread at a time; create one per
 * caller thread.
 *
 * Every function that returns int returns JTK_OK (0) or a negative jtk_status.  The message of the
 * last failure on the calling thread is available from jtk_last_error().
 */
#ifndef JTOKKIT_AMD_H
#define JTOKKIT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes; the comment names the exception the Java shim raises for it. */
typedef enum jtk_status {
    JTK_OK = 0,
    JTK_ERR_INVALID_ARGUMENT = -1,    /* IllegalArgumentException (bad call) */
    JTK_ERR_UNSUPPORTED_SPECIAL = -2, /* UnsupportedOperationException("Encoding special tokens is not
                                         supported yet.")  GptBytePairEncoding.java:52-56 */
    JTK_ERR_UNKNOWN_TOKEN = -3,       /* IllegalArgumentException("Unknown token for decoding: " + id)
                                         GptBytePairEncoding.java:313 */
    JTK_ERR_CAPACITY = -4,            /* caller buffer too small (shim grows and retries) */
    JTK_ERR_BAD_RANK_FILE = -5,       /* IllegalStateException  EncodingFactory.java:142,151,162 */
    JTK_ERR_BAD_UTF8 = -6,            /* input is not what String.getBytes(UTF_8) produces */
    JTK_ERR_NO_DEVICE = -7,           /* no MI355X / HIP runtime: there is NO CPU fallback */
    JTK_ERR_HIP = -8,                 /* a HIP call failed; see jtk_last_error() */
    JTK_ERR_UNSUPPORTED_TABLE = -9,   /* rank table outside what the device path handles (see
                                         jtk_encoding_create) */
    JTK_ERR_PIECE_TOO_LONG = -10,     /* a single unsplittable pre-token piece exceeds JTK_MAX_PIECE_BYTES (1 MiB;
                                         the reference spends O(n^2) on such a piece) */
    JTK_ERR_OUT_OF_MEMORY = -11,
    JTK_ERR_UNENCODABLE = -12         /* IllegalArgumentException("Unknown token for encoding: ...") TokenEncoder.java:66-68: the rank
                                         map lacks a single-byte token and a piece of this document needs it */
} jtk_status;

/* The two split patterns of EncodingFactory.java:63 (= :77, :91) and :105. */
enum { JTK_PATTERN_R50K = 0, JTK_PATTERN_CL100K = 1 };

/* Flags of jtk_batch_encode*. */
enum {
    JTK_ENCODE_ORDINARY = 1u,     /* encodeOrdinary(): skip the special-token check of encode()
                                     (GptBytePairEncoding.java:62-64 vs :47-59) */
    JTK_ENCODE_VALIDATE_UTF8 = 2u,/* also check every document is well-formed UTF-8 (what String.getBytes(UTF_8)
                                     produces); offenders get status JTK_ERR_BAD_UTF8.  Without the flag the
                                     input is trusted: malformed bytes are encoded as the bytes they are. */
    JTK_ENCODE_COUNT_ONLY = 4u,   /* Encoding.countTokens() / countTokensOrdinary() (GptBytePairEncoding.java:122-129) for
                                     the whole batch: token offsets (tok_off[d + 1] - tok_off[d] = the count) and
                                     status, but no token ids -- fetch with tokens == NULL */
    JTK_ENCODE_TO_HOST = 8u       /* jtk_batch_encode only: stream the result to pinned host memory while later chunks
                                     are still being encoded; read it in place with jtk_batch_host_result() */
};

/* Options of jtk_batch_set_option.  A batch larger than one chunk is cut into runs of whole documents ("chunks") that flow
 * through a few scratch sets, each on its own HIP stream: the copies and kernels of consecutive chunks overlap, and the
 * device scratch is sized by the chunk, not by the batch. */
enum {
    JTK_OPT_CHUNK_BYTES = 1,      /* target bytes of text per chunk, device-resident input (default 1 GiB: large chunks
                                     have fewer launches and kernel tails; env JTK_CHUNK_BYTES).  Scratch: ~30 bytes per
                                     byte of chunk per set */
    JTK_OPT_CHUNKS_IN_FLIGHT = 2, /* scratch sets / streams, 1..4 (default 2, and 3 for host input with JTK_ENCODE_TO_HOST unless chosen
                                     here or by env JTK_CHUNKS_IN_FLIGHT: the host waits once per chunk there, two sets stall) */
    JTK_OPT_HOST_CHUNK_BYTES = 3, /* ... host input (default 32 MiB: the copy of one chunk overlaps the kernels of another;
                                     env JTK_HOST_CHUNK_BYTES) */
    JTK_OPT_REUSE_CHUNK_PLAN = 4  /* 1: jtk_batch_encode_device keeps the chunk plan of the last batch while it is handed the same
                                     offsets array again (same pointer and counts: a step loop) -- no plan kernel, no host
                                     synchronisation in the call.  The caller promises not to change those offsets in place
                                     (if it does: JTK_ERR_INVALID_ARGUMENT as the batch's worst status, never wrong tokens).
                                     Default 0 */
};

typedef struct jtk_encoding jtk_encoding;
typedef struct jtk_batch jtk_batch;

const char* jtk_version(void);
const char* jtk_last_error(void);
/* Number of visible HIP devices, or a negative status. */
int jtk_device_count(void);

/* ---- encoding objects -------------------------------------------------------------------------
 * Replaces EncodingFactory.fromPredefinedParameters (:121-137) + the GptBytePairEncoding
 * constructor (GptBytePairEncoding.java:30-35): parses the `.tiktoken` bytes ("base64 SP rank LF",
 * EncodingFactory.java:139-164), builds the device rank tables and uploads them to `device`.
 * `special_literals[i]` / `special_ids[i]` are the special tokens (EncodingFactory.java:24-53).
 *
 * Tables accepted: any rank map with ids < 131071 (JTK_ERR_UNSUPPORTED_TABLE otherwise).  A map that lacks single-byte tokens
 * is taken as the reference takes it: a document with a piece whose merge leaves such a byte alone gets JTK_ERR_UNENCODABLE
 * (TokenEncoder.java:66-68 throws there); it needs one free id above the table's largest per missing byte.  With a token limit
 * the status is conservative: a document is refused when such a piece lies anywhere in the bytes that were encoded (the whole
 * document, or its leading bytes in jtk_batch_encode_max_tokens), where the reference refuses it only if the piece starts
 * before the limit is reached.  The whole-piece lookup of
 * GptBytePairEncoding.java:81-83 is honoured for pieces of any length: for tables in which merging a token's bytes
 * reproduces the token (every table trained by byte-pair merging; the three shipped ones) it is a pure shortcut, for others
 * the unreproducible entries get a lookup of their own and the exact intra-piece cuts are switched off.
 * Special tokens: any number of literals of any length >= 1 (any first byte), ids < 131071 + 2^20. */
int jtk_encoding_create(const char* name, int pattern_kind, const uint8_t* tiktoken, size_t tiktoken_len,
                        const char* const* special_literals, const int32_t* special_ids, int n_specials,
                        int device, jtk_encoding** out);
void jtk_encoding_destroy(jtk_encoding* enc);
const char* jtk_encoding_name(const jtk_encoding* enc);          /* Encoding.getName() */
int jtk_encoding_device(const jtk_encoding* enc);
int64_t jtk_encoding_vocab_size(const jtk_encoding* enc);        /* number of rank-table entries */
int64_t jtk_encoding_pair_count(const jtk_encoding* enc);        /* (left,right) -> rank entries */

/* ---- batch encode: the hot path ------------------------------------------------------------------
 * Replaces a loop of Encoding.encode(String) / encodeOrdinary(String) / countTokens(String)
 * (GptBytePairEncoding.java:38-40, 62-64, 122-129) over n_docs documents.
 *
 * Input: the documents' String.getBytes(UTF_8) bytes back to back in `utf8`, document d occupying
 * [doc_off[d], doc_off[d+1]) (doc_off[0] = 0, doc_off[n_docs] = total bytes).
 * Output (after jtk_batch_fetch / in device memory): token ids of all documents back to back in
 * document order, document d occupying [tok_off[d], tok_off[d+1]); status[d] = JTK_OK or
 * JTK_ERR_UNSUPPORTED_SPECIAL / JTK_ERR_BAD_UTF8 / JTK_ERR_PIECE_TOO_LONG for that document.
 */
int jtk_batch_create(const jtk_encoding* enc, jtk_batch** out);
void jtk_batch_destroy(jtk_batch* b);
int jtk_batch_set_option(jtk_batch* b, int option, int64_t value);

/* Page-locked host memory (hipHostMalloc).  Host buffers handed to jtk_batch_encode are copied by DMA straight from
 * where they are when they were allocated here (what a Java shim does for its direct ByteBuffers); pageable memory
 * works too but is staged by the HIP runtime at a fraction of the link's rate. */
int jtk_host_alloc(size_t bytes, void** out);
void jtk_host_free(void* p);

/* Host buffers: copies the input to the device chunk by chunk (the copy of a chunk overlaps the kernels of the one
 * before), runs the kernels, leaves the result on the device (and, with JTK_ENCODE_TO_HOST, in pinned host memory).
 * *n_tokens receives the total token count (this call synchronises). */
int jtk_batch_encode(jtk_batch* b, const uint8_t* utf8, const int64_t* doc_off, int64_t n_docs,
                     uint32_t flags, int64_t* n_tokens);

/* Custom split patterns (api/GptBytePairEncodingParams.java:36-46: any java.util.regex.Pattern).  Only the two shipped
 * patterns are evaluated on the device; for any other one the caller runs its own matcher on the host and hands over the
 * matches: piece i is utf8[piece_begin[i], piece_end[i]) (positions in the whole batch; ascending, non-empty, not
 * overlapping, each inside one document).  Bytes that no piece covers are not encoded, as `while (matcher.find())`
 * (GptBytePairEncoding.java:79) skips them.  Whole-piece shortcut, bytePairMerge, token order, offsets, status, flags
 * and results are as for jtk_batch_encode (the special-token check of encode() is done on the host here). */
int jtk_batch_encode_pieces(jtk_batch* b, const uint8_t* utf8, const int64_t* doc_off, int64_t n_docs,
                            const int64_t* piece_begin, const int64_t* piece_end, int64_t n_pieces, uint32_t flags, int64_t* n_tokens);

/* After an encode with JTK_ENCODE_TO_HOST: the result in the batch's pinned host buffers (valid until the next encode
 * on this batch): tokens[n_tokens], tok_off[n_docs + 1], status[n_docs]. */
int jtk_batch_host_result(jtk_batch* b, const int32_t** tokens, const int64_t** tok_off, const int32_t** status);

/* Device buffers already resident in HBM (what bench.py times).  `stream_or_null` = a hipStream_t
 * to order against, or NULL for the batch's own stream: the whole encode is ordered like one operation on that stream
 * (inside, the chunks of a large batch run on the batch's own streams, forked from and joined into it).  With
 * n_tokens == NULL the call does not wait for the result (a batch larger than one chunk waits for the work queued on the
 * stream BEFORE it, once, to read the chunk boundaries from d_doc_off); query later with jtk_batch_result().
 * d_utf8 must be 16-byte aligned and readable up to the next multiple of 16 past n_bytes; d_doc_off is checked on the
 * device: offsets that are out of range or decreasing make JTK_ERR_INVALID_ARGUMENT the batch's worst status. */
int jtk_batch_encode_device(jtk_batch* b, const uint8_t* d_utf8, const int64_t* d_doc_off, int64_t n_docs,
                            int64_t n_bytes, uint32_t flags, void* stream_or_null, int64_t* n_tokens);

/* The batch's own HIP stream (a hipStream_t), e.g. to order a caller's work after a non-synchronising encode. */
void* jtk_batch_stream(jtk_batch* b);

/* Synchronises and reports the totals of the last encode. */
int jtk_batch_result(jtk_batch* b, int64_t* n_tokens, int64_t* n_docs, int32_t* worst_status);

/* Copies the last result to host buffers: tokens[tokens_cap] (JTK_ERR_CAPACITY if too small),
 * tok_off[n_docs + 1], status[n_docs]; any of them may be NULL. */
int jtk_batch_fetch(jtk_batch* b, int32_t* tokens, int64_t tokens_cap, int64_t* tok_off, int32_t* status);

/* Device pointers of the last result (valid until the next encode on this batch).  After a host-input job of <= 128 KiB with
 * JTK_ENCODE_TO_HOST they are addresses of pinned host memory the device can reach (the kernels wrote the result there). */
int jtk_batch_device_result(jtk_batch* b, const int32_t** d_tokens, const int64_t** d_tok_off,
                            const int32_t** d_status);

/* Per-kernel device time of the last encode, measured with HIP events on the stream the kernels
 * ran on (enable first; costs a few event records per encode).  names[i] points to static strings. */
int jtk_batch_set_profiling(jtk_batch* b, int enabled);
int jtk_batch_kernel_times(jtk_batch* b, const char** names, float* ms, int cap, int* n);

/* ---- single-document entry points (what the per-call Encoding methods bind) ---------------------
 * Encoding.encode(String[, maxTokens]) / encodeOrdinary(..) (GptBytePairEncoding.java:38-69):
 * max_tokens < 0 means "no limit"; otherwise the result is clipped to max_tokens and backed off to
 * a code-point boundary exactly as :90-100 does, and *truncated gets EncodingResult.isTruncated().
 * utf8 == NULL mirrors text == null (empty result, :48-50). */
int jtk_encode(jtk_batch* b, const uint8_t* utf8, int64_t len, uint32_t flags, int64_t max_tokens,
               int32_t* tokens, int64_t tokens_cap, int64_t* n_tokens, int* truncated);

/* Encoding.decodeBytes(List<Integer>) (GptBytePairEncoding.java:137-151, 302-314): concatenates the
 * byte strings of `ids`; *len receives the byte count (out may be NULL to size). */
int jtk_decode(const jtk_encoding* enc, const int32_t* ids, int64_t n, uint8_t* out, int64_t cap, int64_t* len);

/* ---- maxTokens for a whole batch, on the device ------------------------------------------------------
 * Encoding.encode(String, int maxTokens) / encodeOrdinary(String, int) (GptBytePairEncoding.java:43-45, 66-69) for
 * every document of the LAST batch encode on `b`: document d keeps the first kept[d] of its tokens
 * ([tok_off[d], tok_off[d] + kept[d]) of the encode result) -- min(maxTokens, count) backed off to a code-point
 * boundary exactly as :90-100 does -- and truncated[d] = EncodingResult.isTruncated() (:97).
 * The input text of that encode must still be where it was (host-buffer encodes keep their own copy). */
int jtk_batch_truncate(jtk_batch* b, int64_t max_tokens);
int jtk_batch_fetch_truncated(jtk_batch* b, int64_t* kept, uint8_t* truncated);           /* [n_docs] each, may be NULL */
int jtk_batch_device_truncated(jtk_batch* b, const int64_t** d_kept, const uint8_t** d_truncated);

/* Encoding.encode(text, maxTokens) / encodeOrdinary(text, maxTokens) for every document WITHOUT encoding the documents whole
 * (the reference stops matching at maxTokens, GptBytePairEncoding.java:83-88): leading bytes of each document are encoded
 * (8 per wanted token + 64, 4x more for the documents that turn out to need it) and a result is taken once it is certain to be the
 * head of the full token list.  Host buffers.  tokens: [n_docs * max_tokens], document d's ids at tokens + d * max_tokens;
 * kept[d] ids are valid; truncated[d] (may be NULL) = EncodingResult.isTruncated(); status[d] (may be NULL) is JTK_OK or
 * JTK_ERR_UNSUPPORTED_SPECIAL.  flags: JTK_ENCODE_ORDINARY or 0. */
int jtk_batch_encode_max_tokens(jtk_batch* b, const uint8_t* utf8, const int64_t* doc_off, int64_t n_docs, uint32_t flags,
                                int64_t max_tokens, int32_t* tokens, int64_t* kept, uint8_t* truncated, int32_t* status);

/* ---- batch decode on the device ---------------------------------------------------------------------
 * Replaces a loop of Encoding.decodeBytes(List<Integer>) (GptBytePairEncoding.java:137-151, 302-314; special-token
 * ids decode to their literals, :308-311) over n_seqs token lists: all ids back to back in `ids`, list q occupying
 * [seq_off[q], seq_off[q+1]).  Result: the byte strings back to back, list q occupying [byte_off[q], byte_off[q+1]);
 * status[q] = JTK_OK or JTK_ERR_UNKNOWN_TOKEN (that list's bytes then omit the unknown ids).  Both calls synchronise
 * and leave the result on the device; *n_bytes receives the total byte count. */
int jtk_batch_decode(jtk_batch* b, const int32_t* ids, const int64_t* seq_off, int64_t n_seqs, int64_t* n_bytes);
int jtk_batch_decode_device(jtk_batch* b, const int32_t* d_ids, const int64_t* d_seq_off, int64_t n_seqs, int64_t n_ids,
                            void* stream_or_null, int64_t* n_bytes);
/* Copies the last decode to host buffers: out[out_cap] (JTK_ERR_CAPACITY if too small), byte_off[n_seqs + 1],
 * status[n_seqs]; any may be NULL. */
int jtk_batch_decode_fetch(jtk_batch* b, uint8_t* out, int64_t out_cap, int64_t* byte_off, int32_t* status);
/* Device pointers of the last decode (valid until the next decode on this batch). */
int jtk_batch_decode_device_result(jtk_batch* b, const uint8_t** d_out, const int64_t** d_byte_off, const int32_t** d_status);

/* ---- per-call service: many caller threads, one device batch at a time -------------------------------------------
 * The reference is called per document from many threads (api/Encoding.java; its benchmark is one task per document on a
 * pool of 1..64 threads, benchmark/.../AbstractMultiThreadedBenchmark.java:35-45).  A jtk_service coalesces such callers:
 * whatever is queued when a worker becomes free is encoded as ONE device batch (no timer; while a batch is on the device
 * the next one piles up), and every caller gets its own tokens back.  Thread-safe.  n_workers (default 2) worker threads,
 * each with its own jtk_batch, so that the gathering of one batch overlaps the device time of another.
 *   jtk_service_encode   blocking: Encoding.encode / encodeOrdinary / countTokens (flags as for jtk_batch_encode;
 *                        max_tokens < 0 = no limit) -- same results and status codes as jtk_encode
 *   jtk_service_submit / jtk_service_wait   the same in two halves, so that one thread can keep many documents in flight;
 *                        utf8 and tokens must stay valid until the wait returns; every ticket must be waited for once */
typedef struct jtk_service jtk_service;
typedef struct jtk_ticket jtk_ticket;
int jtk_service_create(const jtk_encoding* enc, int n_workers, jtk_service** out);
void jtk_service_destroy(jtk_service* s);
int jtk_service_encode(jtk_service* s, const uint8_t* utf8, int64_t len, uint32_t flags, int64_t max_tokens,
                       int32_t* tokens, int64_t tokens_cap, int64_t* n_tokens, int* truncated);
int jtk_service_submit(jtk_service* s, const uint8_t* utf8, int64_t len, uint32_t flags, int64_t max_tokens,
                       int32_t* tokens, int64_t tokens_cap, jtk_ticket** ticket);
int jtk_service_wait(jtk_service* s, jtk_ticket* ticket, int64_t* n_tokens, int* truncated);
int jtk_service_done(const jtk_ticket* ticket);     /* 1: the result is in (jtk_service_wait returns at once), 0: not yet -- a poll for callers that must not block */
int jtk_service_stats(jtk_service* s, int64_t* n_batches, int64_t* n_docs);     /* device batches run, documents encoded */
/* What one device batch takes from the queue at most (defaults: 65536 documents, 64 MiB of text; a single larger document still
 * goes alone); the rest stays queued for the next batch.  Bounds the batch's pinned staging.  Values < 1 leave a limit as it is. */
int jtk_service_set_limits(jtk_service* s, int64_t max_docs, int64_t max_bytes);

/* ---- multi-GPU: document shards and the offset stitch ------------------------------------------------------------
 * Documents are independent (every Encoding.encode call is a pure function of one string, GptBytePairEncoding.java:71-103),
 * so a batch shards as contiguous document ranges balanced by bytes, one per GPU / process, each process with its own
 * jtk_encoding (the rank tables are a few MB) and jtk_batch.  The one exchange step is an RCCL all-gather of the per-shard
 * token totals (1 x int64 per rank, over xGMI); the exclusive prefix is the shard's first global token.  RCCL is bound at
 * run time (librccl.so; env JTK_RCCL_LIB), only when a communicator is created. */
typedef struct jtk_comm jtk_comm;

/* bounds[world + 1]: rank r encodes documents [bounds[r], bounds[r + 1]) (host arrays; doc_off has n_docs + 1 entries). */
int jtk_shard_plan(const int64_t* doc_off, int64_t n_docs, int world, int64_t* bounds);
/* ncclGetUniqueId: call on one rank, hand the 128 bytes to every rank by any side channel, then every rank creates its
 * communicator (ncclCommInitRank; collective over the ranks). */
int jtk_comm_unique_id(uint8_t* id128);
int jtk_comm_create(const uint8_t* id128, int world, int rank, int device, jtk_comm** out);
void jtk_comm_destroy(jtk_comm* c);
int jtk_comm_world(const jtk_comm* c);
int jtk_comm_rank(const jtk_comm* c);
/* The stitch, queued on `stream` (e.g. jtk_batch_stream() right after a non-synchronising jtk_batch_encode_device; nothing
 * waits for the host): all-gather of d_tok_off[n_docs] (this shard's token total), base = sum of the lower ranks' totals,
 * d_global_off[d] = d_tok_off[d] + base for d = 0..n_docs (d_global_off may be NULL: totals and base only).
 * *d_totals (world entries) and *d_base (1 entry) are the communicator's device buffers. */
int jtk_comm_stitch(jtk_comm* c, const int64_t* d_tok_off, int64_t n_docs, int64_t* d_global_off, void* stream,
                    const int64_t** d_totals, const int64_t** d_base);
/* Synchronises `stream` and copies the last stitch's totals[world] and base to the host. */
int jtk_comm_fetch(jtk_comm* c, void* stream, int64_t* totals, int64_t* base);

#define JTK_MAX_PIECE_BYTES (1 << 20)

#ifdef __cplusplus
}
#endif
#endif /* JTOKKIT_AMD_H */
