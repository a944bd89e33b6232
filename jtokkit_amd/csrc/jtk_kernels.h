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
// atomic per tile and bin.  Entry: pos (40 bits) | len << 40 (10 bits) | tokens in its tile << 50 (set by the merge).
#define JTK_NBINS 5
#define JTK_Q_SHARDS 64
#define JTK_BIN_CAP0 (JTK_TILE / 2)    // per tile: pieces of 2..16 bytes
#define JTK_BIN_CAP1 (JTK_TILE / 16)   //           17..32 bytes
#define JTK_BIN_CAP2 (JTK_TILE / 32)   //           33..64 bytes
#define JTK_BIN_CAP3 (JTK_TILE / 64)   //           65..128 bytes
#define JTK_BIN_CAP4 (JTK_TILE / 128)  //           129..256 bytes
#define JTK_BIN_MAXLEN 256            // longer pieces go to the wave-per-piece kernels
#define JTK_M_WGS_PER_SHARD 4
#define JTK_MID_CAP 512          // wave-per-piece kernel, small bin: pieces of 65..512 bytes
#define JTK_LONG_CAP 8192        // wave-per-piece kernel, large bin
#define JTK_GIANT_CAP (1 << 20)  // workgroup-per-piece kernel with parts in global scratch (= JTK_MAX_PIECE_BYTES)
#define JTK_GIANT_CHUNK 256      // positions per cached chunk minimum
#define JTK_MAX_SPECIALS 8
#define JTK_SPECIAL_MAXLEN 32

struct JtkDeviceTables {
    JtkUcTables uc;
    const uint32_t* byte_rank;   // [256]
    JtkPairTable pairs;
    JtkTok8Table tok8;
    const uint32_t* bp_rank;     // [65536]
    JtkBpLds bp;                 // the same, compressed (staged into LDS by bpe_merge)
    const uint32_t* pair_in_token;   // [2048] bit (b0 << 8 | b1): adjacent inside some table entry
    int kind;
    int n_specials;
    uint8_t special_len[JTK_MAX_SPECIALS];
    uint8_t special[JTK_MAX_SPECIALS][JTK_SPECIAL_MAXLEN];
};

struct JtkLongPiece {
    int64_t start;
    int64_t len;
};

struct JtkResult {
    int64_t n_tokens;
    int32_t worst_status;
    uint32_t n_giant;       // pieces longer than JTK_LONG_CAP, handled in a second phase after a host check
};

// Device-side working set of one encode call (all pointers into the batch's scratch).
struct JtkWork {
    const uint8_t* text;
    const int64_t* doc_off;
    int64_t n_bytes;
    int64_t n_docs;
    int64_t n_words;        // 64-bit mask words (covers position n_bytes, plus padding)
    int64_t n_tiles;
    uint64_t* docmask;      // bit p: a document starts at byte p
    uint64_t* piecemask;    // bit p: a pre-token piece starts at byte p (bit n_bytes is a sentinel)
    uint64_t* tokmask;      // bit p: a token starts at byte p
    uint16_t* blk_pre;      // per 64-byte block: tokens of its tile before the block
    uint32_t* tok_at;       // per byte position: id of the token starting there, or JTK_ID_DEAD
    uint32_t* tile_cnt;     // tokens starting in each tile
    int64_t* tile_off;      // exclusive scan of tile_cnt (n_tiles + 1)
    uint64_t* q[JTK_NBINS];         // [JTK_Q_SHARDS][q_cap[k]] queue entries of bin k
    int64_t q_cap[JTK_NBINS];       // entries per shard
    uint32_t* q_count;              // [JTK_NBINS][JTK_Q_SHARDS]
    uint32_t* q_base[JTK_NBINS];    // [n_tiles] where in its shard a tile's entries of bin k start ...
    uint32_t* q_n[JTK_NBINS];       // [n_tiles] ... and how many there are
    JtkLongPiece* mid_list; // pieces of 65..JTK_MID_CAP bytes
    JtkLongPiece* long_list;// longer pieces
    JtkLongPiece* giant_list;// pieces longer than JTK_LONG_CAP (second phase)
    uint32_t* mid_count;
    uint32_t* long_count;
    int32_t* status;        // per document
    int32_t* tokens;        // output, packed
    int64_t* tok_off;       // output, n_docs + 1
    JtkResult* result;
};

void jtk_launch_mark_docs(const JtkWork& w, hipStream_t s);
void jtk_launch_special_check(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s);
void jtk_launch_validate_utf8(const JtkWork& w, hipStream_t s);
void jtk_launch_pretok_split(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s);
void jtk_launch_piece_resolve(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s);
void jtk_launch_bpe_merge(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s);
void jtk_launch_bpe_merge_long(const JtkWork& w, const JtkDeviceTables& t, hipStream_t s);
void jtk_launch_bpe_merge_giant(const JtkWork& w, const JtkDeviceTables& t, uint32_t n_giant, const int64_t* scratch_off,
                                uint32_t* scratch, hipStream_t s);
void jtk_launch_tile_scan(const JtkWork& w, hipStream_t s);
void jtk_launch_pack(const JtkWork& w, hipStream_t s);

#endif
