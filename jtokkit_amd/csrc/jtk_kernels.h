// jtk_kernels.h -- launch interface of the gfx950 kernels (implemented in jtk_kernels.hip).
#ifndef JTK_KERNELS_H
#define JTK_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jtk_common.h"

#define JTK_SPLIT_TILE 4096      // bytes per pretok_split workgroup
#define JTK_SPLIT_HALO 64
#define JTK_TILE 2048            // bytes per piece_resolve / pack workgroup; token counts are kept per tile
// Pieces that need bytePairMerge are queued by length bin; bin k holds pieces of up to JTK_BIN_SLOTS(k) bytes.
// Queues are dense and sharded: tile t appends its entries to shard t % JTK_Q_SHARDS with one returning
// atomic per tile and bin.  A queue entry is two words in two parallel arrays:
//   qm  (8 bytes)   pos (37 bits) | (len - 1) << 37
//   qd  (16 bytes)  bins 0..2: IN the piece's bytes (<= 16, what piece_resolve hashed), OUT the merge result;
//                   bins 3..6: OUT the merge result (their bytes are read from the text)
// Merge result: (token count - 1) in the top byte | up to seven token ids, 17 bits each from bit 0; a piece that
// became more than seven tokens leaves its tokens in htok (packed from its first byte position).  The merge kernels
// add every piece's token count to tile_tot[tile] (piece_resolve stored the resolved pieces' count there).
#define JTK_QE_POS_MASK ((1ull << 37) - 1ull)
#define JTK_QE_LEN_SHIFT 37
#define JTK_QE_DONE (1ull << 63)       // the piece is a table entry found by k_long_shortcut: its result is in place already
#define JTK_NBINS 7
// Pieces of 2 or 3 bytes that are not table entries need no pair-table lookup at all (at most one merge of a 2-byte token;
// the pair that would follow is the whole piece, which is not an entry).  They get a queue of their own ("bin" JTK_BIN_TINY
// of a piece-list entry) with 8-byte entries that carry the bytes: pos (37 bits) | (len - 2) << 37 | b0 << 40 | b1 << 48 |
// b2 << 56; the merge kernel replaces an entry by its result: up to three token ids, 17 bits each from bit 0 (the first 64
// bits of a merge result word), | (count - 1) << 62.
#define JTK_BIN_TINY 7
#define JTK_TINY_CAP (JTK_TILE / 2)
#define JTK_Q_SHARDS 64
// Every queue counter has a 128-byte line of its own: the tiles' claims are returning atomics, and atomics on one line
// are served one after another (about 14 ns each): with the 512 counters packed into 16 lines the claims of a GiB of mixed
// text -- one per tile and bin -- took 1.8 ms, most of piece_resolve's time.
#define JTK_QC_STRIDE 32
#define JTK_QC(bin, shard) (((bin) * JTK_Q_SHARDS + (shard)) * JTK_QC_STRIDE)
// Bins by length.  A wave of the merge kernel steps until its longest piece is done, so pieces of like length are queued
// together: three classes up to 16 bytes (whose entries carry the piece's bytes), then powers of two.
#define JTK_BIN_CAP0 (JTK_TILE / 4)    // per tile: pieces of 4..8 bytes (2..3-byte pieces: JTK_BIN_TINY)
#define JTK_BIN_CAP1 (JTK_TILE / 8)    //           9..12 bytes
#define JTK_BIN_CAP2 160               //           13..16 bytes (2048 / 13 = 157)
#define JTK_BIN_CAP3 (JTK_TILE / 16)   //           17..32 bytes
#define JTK_BIN_CAP4 (JTK_TILE / 32)   //           33..64 bytes
#define JTK_BIN_CAP5 (JTK_TILE / 64)   //           65..128 bytes
#define JTK_BIN_CAP6 (JTK_TILE / 128)  //           129..256 bytes
// the head of a tile's slice of each bin's queue that pack stages in LDS: 32 + 16 + 16 results of the three classes of <= 16
// bytes (staging slots 0..63), 8 of each longer bin (64..95); tiny pieces have a staging area of their own (JTK_PACK_TINY).
// (Three times as much -- enough for every tile of CJK text -- made pack 3 % faster on mixed text and 10 % slower on prose.)
#define JTK_PACK_TINY 128
#define JTK_PACK_SLOTS 96
#define JTK_PACK_CAP(bin) ((bin) == 0 ? 32 : (bin) <= 2 ? 16 : 8)
#define JTK_PACK_OFF(bin) ((bin) == 0 ? 0 : (bin) == 1 ? 32 : (bin) == 2 ? 48 : 64 + ((bin) - 3) * 8)
#define JTK_NBINS_BYTES 3              // bins 0..2: the queue entry carries the piece's bytes
#define JTK_NBINS_LEAN 5               // bins 0..4: lean phases of the merge kernel; 5..6: state-machine phases
#define JTK_BIN_MAXLEN 256            // longer pieces go to the wave-per-piece kernels
#define JTK_M_WGS_PER_SHARD 4
#define JTK_MID_CAP 512          // wave-per-piece kernel, small bin: pieces of 65..512 bytes
#define JTK_LONG_CAP 8192        // wave-per-piece kernel, large bin
#define JTK_GIANT_CAP (1 << 20)  // workgroup-per-piece kernel with parts in global scratch (= JTK_MAX_PIECE_BYTES)
#define JTK_GIANT_CHUNK 256      // positions per cached chunk minimum
#define JTK_MAX_SPECIALS 65536           // special-token literals live in a device blob: the bounds are sanity checks only
#define JTK_SPECIAL_MAXLEN 65535

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
    uint32_t pseudo_base;            // ids from here on stand for single bytes that are no tokens (0: the table has all 256)
    int n_specials;
    uint32_t special_first[8];       // bit b: some special-token literal starts with byte b
    const uint8_t* special_blob;     // the literals back to back
    const uint32_t* special_off;     // [n_specials + 1] into special_blob
};

// piece-list entry.  Resolved piece: token id (bits 0..16) | byte offset in the tile << 17.
// Merged piece: JTK_PL_HARD | byte offset in the tile (bits 0..10) and either its queue entry (bin << 21 | index in
// the tile's slice of the bin's queue << 11) or JTK_PL_NOQUEUE (wave / workgroup kernels: tokens and count in htok).
#define JTK_PL_HARD 0x80000000u
#define JTK_PL_NOQUEUE 0x40000000u
#define JTK_PL_STAGED 0x20000000u        // queued piece whose result is in the head of the tile's queue slice that pack stages in LDS:
                                         // the index field then holds its staging slot (pack_stage_slot), not the index in the bin
#define JTK_PL_OFF_SHIFT 17
#define JTK_PL_QI_SHIFT 11
#define JTK_PL_BIN_SHIFT 21
#define JTK_HT_ID_MASK 0x1FFFFu
#define JTK_HT_CNT_SHIFT 17
#define JTK_HT_CNT_MASK 0x3FFFu
#define JTK_HT_ESCAPE 0x3FFFu            // count does not fit: giant piece, count in docpre[pos + 1]

struct JtkLongPiece {
    int64_t start;
    int64_t len;
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
    int64_t n_tiles;
    uint32_t count_only;    // countTokens(): pack computes the offsets but writes no token ids
    uint32_t inline_scan;   // small single-chunk job: pack adds up the tiles before its own itself and k_tile_scan is not launched
    uint32_t check_special; // encode(): flag documents that contain a special-token literal (done inside pretok_split)
    uint64_t* docmask;      // bit p: a document starts at byte p
    uint64_t* piecemask;    // bit p: a pre-token piece starts at byte p (bit n_bytes is a sentinel)
    uint64_t* gapmask;      // NULL, or (caller-supplied pieces, jtk_batch_encode_pieces) bit p: the "piece" that starts at byte p is
                            // text between two matches of the caller's pattern: it is not encoded (matcher.find() skips it)
    uint32_t* plist;        // [n_tiles * JTK_TILE] per tile, packed from the tile's first word: its pieces in text order,
                            // JTK_PL_* entry per piece (a piece belongs to the tile it starts in)
    uint32_t* tile_np;      // [n_tiles] pieces in each tile's list
    uint32_t* htok;         // [n_tiles * JTK_TILE] tokens of a merged piece without a (big enough) result slot, packed from the
                            // piece's first byte position (k <= len words); word 0 also carries the count k: id | k << 17
                            // (JTK_HT_ESCAPE: the count is in docpre[pos + 1])
    uint32_t* docpre;       // [n_tiles * JTK_TILE] at a document's first byte: tokens of its tile before it (sparse)
    uint32_t* tile_tot;     // [n_tiles] tokens of the tile's pieces: piece_resolve stores the resolved pieces (one token
                            // each), the merge kernels add theirs
    int64_t* tile_off;      // [n_tiles + 1] exclusive scan of tile_tot
    uint64_t* qm[JTK_NBINS];        // [JTK_Q_SHARDS][q_cap[k]] queue entries of bin k: position and length
    uint4* qd[JTK_NBINS];           // [JTK_Q_SHARDS][q_cap[k]] ... : bytes in (bin 0), merge result out
    int64_t q_cap[JTK_NBINS];       // entries per shard
    uint64_t* qt;                   // [JTK_Q_SHARDS][qt_cap] the queue of JTK_BIN_TINY: bytes in, result out
    int64_t qt_cap;
    uint32_t* q_count;              // [JTK_NBINS + 1][JTK_Q_SHARDS], one counter per 128-byte line: JTK_QC(bin, shard)
    uint32_t* q_meta;               // [n_tiles][16]: [k] where in its shard the tile's entries of bin k start, [8 + k] how many
    JtkLongPiece* mid_list; // pieces of 65..JTK_MID_CAP bytes
    JtkLongPiece* long_list;// longer pieces
    JtkLongPiece* giant_list;// pieces longer than JTK_LONG_CAP
    uint32_t* mid_count;
    uint32_t* long_count;
    uint32_t* n_giant;      // pieces longer than JTK_LONG_CAP (listed by piece_resolve, merged by the last phase of k_bpe_merge_all)
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
void jtk_launch_piece_resolve(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s);
void jtk_launch_long_shortcut(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s);     // only if t.longtok.n
void jtk_launch_bpe_merge(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s);
void jtk_launch_tile_scan(const JtkWork& w, hipStream_t s);
void jtk_launch_pack(const JtkWork& w, hipStream_t s);
void jtk_launch_doc_offsets(const JtkWork& w, hipStream_t s);
void jtk_launch_flag_unencodable(const JtkWork& w, uint32_t pseudo_base, hipStream_t s);

#endif
