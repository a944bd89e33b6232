// jtk_kernels.h -- launch interface of the gfx950 kernels (implemented in jtk_kernels.hip).
#ifndef JTK_KERNELS_H
#define JTK_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jtk_common.h"

#define JTK_SPLIT_TILE 4096      // bytes per pretok_split workgroup
#define JTK_SPLIT_HALO 64
// A STRIP is the unit of the encode kernel: 4096 bytes = 64 piece-mask words, one per lane of the wave that encodes it.
// Token counts, the scan and the per-document offsets are kept per strip ("tile" in the names below).
#define JTK_TILE 4096
// Pieces that need bytePairMerge are queued by k_strip_encode for k_bpe_merge, by length bin, in dense queues, one per bin and
// SHARD = workgroup of k_strip_encode (which claims its entries with an atomic in LDS: no global atomic on this path -- a
// few counters shared by the whole grid took 1 ns per claim, 5 ms per GiB of mixed text).  An entry is 16 bytes:
//   x, y  pos (37 bits) | (len - 1) << 37   (JTK_QE_DONE: found by k_long_shortcut, nothing left to merge)
//   z, w  index of the piece's slot and hole record (stok / hrec)
// Bins 0..2 are pieces of up to JTK_SHORT_MAX bytes that are no table entry (the whole-piece lookup was made by
// k_strip_encode), bins 3..6 longer ones (for which the rank table's entries of more than 16 bytes are found by merging).
#define JTK_SHORT_MAX 16
#define JTK_QE_POS_MASK ((1ull << 37) - 1ull)
#define JTK_QE_LEN_SHIFT 37
#define JTK_QE_DONE (1ull << 63)
#define JTK_NBINS 7                    // queue bins: pieces of 4..8, 9..12, 13..16, 17..32, ..64, ..128, ..256 bytes
#define JTK_NBINS_SHORT 3
#define JTK_MAX_Q_SHARDS 1024
#define JTK_BIN_CAP0 (JTK_TILE / 4)    // per strip: pieces of 4..8 bytes (2..3-byte pieces never need the pair table)
#define JTK_BIN_CAP1 (JTK_TILE / 8)    //            9..12 bytes
#define JTK_BIN_CAP2 320               //            13..16 bytes (4096 / 13 = 315)
#define JTK_BIN_CAP3 (JTK_TILE / 16)   //            17..32 bytes
#define JTK_BIN_CAP4 (JTK_TILE / 32)   //            33..64 bytes
#define JTK_BIN_CAP5 (JTK_TILE / 64)   //            65..128 bytes
#define JTK_BIN_CAP6 (JTK_TILE / 128)  //            129..256 bytes
#define JTK_BIN_MAXLEN 256            // longer pieces go to the wave-per-piece phases
#define JTK_M_WGS_PER_SHARD 4
#define JTK_MID_CAP 512          // wave-per-piece phase, small bin: pieces of 257..512 bytes
#define JTK_LONG_CAP 8192        // wave-per-piece phase, large bin
#define JTK_GIANT_CAP (1 << 20)  // workgroup-per-piece phase with parts in global scratch (= JTK_MAX_PIECE_BYTES)
#define JTK_GIANT_CHUNK 256      // positions per cached chunk minimum
#define JTK_MAX_SPECIALS 8
#define JTK_SPECIAL_MAXLEN 32

#define JTK_UC_LDS_STAGE1 4352    // capacity of the LDS copy of the Unicode class table (pretok_split)
#define JTK_UC_LDS_STAGE2 2048

struct JtkDeviceTables {
    JtkUcTables uc;
    uint32_t uc_stage1_len, uc_stage2_words;
    const uint32_t* byte_rank;   // [256]
    JtkPairTable pairs;
    JtkTok8Table tok8;
    JtkTok16Table tok16;
    const uint32_t* bp_rank;     // [65536]
    JtkBpLds bp;                 // the same, compressed (staged into LDS by bpe_merge)
    const uint32_t* pair_in_token;   // [2048] bit (b0 << 8 | b1): adjacent inside some table entry
    uint32_t lead_letters[8];        // bit b: every character whose UTF-8 form starts with byte b is a letter (jtk_lead_all_letters)
    JtkLongTokTable longtok;         // table entries of > 16 bytes that merging does not reproduce (n == 0 for the shipped tables)
    int kind;
    int n_specials;
    uint8_t special_len[JTK_MAX_SPECIALS];
    uint8_t special[JTK_MAX_SPECIALS][JTK_SPECIAL_MAXLEN];
};

#define JTK_HT_ID_MASK 0x1FFFFu

struct JtkLongPiece {       // a piece of more than JTK_BIN_MAXLEN bytes
    int64_t start;
    uint64_t idx_len;       // index of its slot and hole record (40 bits) | length << 40   (length 0: found by k_long_shortcut)
};

struct JtkResult {          // of a whole batch (all its chunks)
    int64_t n_tokens;
    int32_t worst_status;
    uint32_t pad;
};

// Device-side working set of ONE CHUNK of an encode call: a run of whole documents of the batch, encoded with one scratch
// set.  A batch is one chunk, or several that flow through a few scratch sets on their own streams (jtk_abi.cpp).
// Positions are relative to the chunk's origin `text` = batch text + text_base (text_base is a multiple of JTK_TILE, so the
// chunk's first document starts `lead` < JTK_TILE bytes in; the bytes before it belong to the previous chunk and start no
// piece here).  doc_off / status / tok_off point at the chunk's first document in the batch-wide arrays.
struct JtkWork {
    const uint8_t* text;
    const int64_t* doc_off; // [n_docs + 1] positions in the whole batch (subtract text_base)
    int64_t text_base;
    int64_t lead;
    int64_t n_bytes;        // from the chunk's origin to the end of its last document
    int64_t n_docs;
    int64_t n_words;        // 64-bit mask words (covers position n_bytes, plus padding)
    int64_t n_tiles;        // strips
    uint32_t count_only;    // countTokens(): the offsets are computed but no token ids are written
    uint32_t inline_scan;   // small single-chunk job: the expand kernel adds up the strips before its own itself (no k_tile_scan)
    uint32_t check_special; // encode(): flag documents that contain a special-token literal (done inside pretok_split)
    uint64_t* docmask;      // bit p: a document starts at byte p
    uint64_t* piecemask;    // bit p: a pre-token piece starts at byte p (bit n_bytes is a sentinel)
    uint64_t* gapmask;      // NULL, or (caller-supplied pieces, jtk_batch_encode_pieces) bit p: the "piece" that starts at byte p is
                            // text between two matches of the caller's pattern: it is not encoded (matcher.find() skips it)
    uint32_t* stok;         // [n_tiles * JTK_TILE] per strip, one slot per piece in text order: the token of a DENSE piece (<= 8 bytes,
                            // found in its primary tok8 slot), or SLOT_HOLE
    uint64_t* hrec;         // [n_tiles * JTK_TILE] per strip, by hole number: the hole's tokens (HR_* in jtk_strip_encode.h)
    uint32_t* tile_np;      // [n_tiles] pieces of the strip
    uint32_t* sbase;        // [n_tiles] where the strip's slots start in its wave's region of stok / hrec (jtk_strip_common.h)
    uint32_t wave_cap;      // indices per wave region
    uint32_t enc_waves;     // waves per workgroup of k_strip_encode
    uint4* memo;            // NULL, or [8 XCDs][memo_mask + 1][2]: merged pieces remembered for the rest of the job (jtk_strip_encode.h)
    uint32_t memo_mask;
    uint32_t* htok;         // [n_tiles * JTK_TILE] tokens of a merged piece that became more than three tokens, packed from the
                            // piece's first byte position (k <= len words; their hole record says how many)
    uint32_t* docpre;       // [n_tiles * JTK_TILE] at a document's first byte: tokens of its strip before it (sparse)
    uint32_t* tile_tot;     // [n_tiles] tokens of the strip (zeroed per job; k_strip_encode adds to it)
    int64_t* tile_off;      // [n_tiles + 1] exclusive scan of tile_tot
    uint4* qe[JTK_NBINS];           // [n_shards][q_cap[k]] queue entries of bin k
    int64_t q_cap[JTK_NBINS];       // entries per shard
    uint32_t* q_count;              // [JTK_NBINS][n_shards] (written by each workgroup of k_strip_encode when it is done)
    uint32_t n_shards;              // = the grid of k_strip_encode
    JtkLongPiece* mid_list; // pieces of 257..JTK_MID_CAP bytes
    JtkLongPiece* long_list;// longer pieces
    JtkLongPiece* giant_list;// pieces longer than JTK_LONG_CAP
    uint32_t* mid_count;
    uint32_t* long_count;
    uint32_t* n_giant;      // pieces longer than JTK_LONG_CAP (merged by the last phase of k_bpe_merge)
    int32_t* status;        // per document
    int32_t* tokens;        // output of the whole batch, packed (tile_off already includes the earlier chunks' tokens)
    int64_t* tok_off;       // output, n_docs + 1
    const int64_t* job_tokens;   // tokens of the batch's earlier chunks (written by the previous chunk's scan)
    int64_t* job_tokens_next;    // ... including this one (for the next chunk)
    int64_t* set_info;      // [2] (device-visible host memory) first token of this chunk, end of its last
    JtkResult* result;
};

// Device-side working set of one batch decode (jtk_decode.hip).
#define JTK_DEC_TILE 2048        // tokens per decode workgroup
struct JtkDecodeWork {
    const int32_t* ids;         // all sequences' token ids back to back
    const int64_t* seq_off;     // [n_seqs + 1]
    int64_t n_tok, n_seqs, n_tiles;
    const uint32_t* tab_off;    // [n_ids_table + 1] byte offset of every id's byte string in tab_blob (absent id: empty)
    const uint8_t* tab_blob;
    uint32_t n_ids_table;
    uint64_t* seqmask;          // bit t: a sequence starts at token t (zeroed per call)
    uint32_t* tile_bytes;       // [n_tiles]
    int64_t* tile_off;          // [n_tiles + 1]
    uint32_t* seqpre;           // at a sequence's first token: bytes of its tile before it (sparse)
    int32_t* status;            // [n_seqs] (zeroed per call)
    int32_t* worst_status;
    int64_t* total;
    uint8_t* out;               // NULL in the sizing phase
    int64_t* byte_off;          // [n_seqs + 1]
};
struct JtkTruncWork {
    const int32_t* tokens;      // result of the last batch encode
    const int64_t* tok_off;
    const uint8_t* text;
    const int64_t* doc_off;
    int64_t n_docs;
    const uint32_t* tab_off;    // decode table offsets (token byte lengths)
    int64_t max_tokens;
    int64_t* kept;              // [n_docs] tokens kept per document
    uint8_t* truncated;         // [n_docs] EncodingResult.isTruncated()
};
void jtk_launch_truncate(const JtkTruncWork& w, hipStream_t s);
void jtk_launch_decode_count(const JtkDecodeWork& w, hipStream_t s);     // mark, count, scan
void jtk_launch_decode_scatter(const JtkDecodeWork& w, hipStream_t s);   // scatter, offsets

// chunk plan of a batch whose offsets are in device memory: out_doc[c], out_off[c] for c = 0..n_chunks
void jtk_launch_plan_chunks(const int64_t* doc_off, int64_t n_docs, int64_t chunk_bytes, int n_chunks, int64_t* out_doc, int64_t* out_off,
                            hipStream_t s);
void jtk_launch_stitch(const int64_t* totals, int rank, int64_t* base_out, const int64_t* tok_off, int64_t n_docs, int64_t* global_off,
                       hipStream_t s);
void jtk_launch_mark_docs(const JtkWork& w, hipStream_t s);
// caller-supplied pieces [begin[i], end[i]) (positions in the whole batch), i = 0..n_pieces-1, instead of pretok_split
void jtk_launch_mark_pieces(const JtkWork& w, const int64_t* begin, const int64_t* end, int64_t n_pieces, hipStream_t s);
void jtk_launch_validate_utf8(const JtkWork& w, hipStream_t s);
void jtk_launch_pretok_split(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s);
int jtk_strip_encode_grid(int64_t n_tiles);                                                   // workgroups of k_strip_encode = queue shards
int jtk_strip_encode_waves(void);                                                             // waves per workgroup
void jtk_launch_strip_encode(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s);      // every strip: dense tokens, hole bitmap, hole records; queues
void jtk_launch_long_shortcut(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s);     // only if t.longtok.n
void jtk_launch_bpe_merge(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s);         // the queued pieces' hole records
void jtk_launch_tile_scan(const JtkWork& w, hipStream_t s);
void jtk_launch_strip_expand(const JtkWork& w, hipStream_t s);                                // ... -> tokens, docpre
void jtk_launch_doc_offsets(const JtkWork& w, hipStream_t s);

#endif
