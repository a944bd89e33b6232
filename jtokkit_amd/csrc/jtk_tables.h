// jtk_tables.h -- host-side rank tables of one encoding (parsed once, then uploaded).
#ifndef JTK_TABLES_H
#define JTK_TABLES_H

#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "jtk_common.h"

struct JtkHostTables {
    std::string name;
    int kind = 0;
    // TokenEncoder (TokenEncoder.java:16-17): both directions
    std::unordered_map<std::string, uint32_t> bytes_to_id;
    std::vector<std::string> id_to_bytes;        // "" = id absent (p50k has a hole at 50256)
    std::vector<uint8_t> id_present;
    std::vector<std::pair<std::string, int32_t>> specials;
    uint32_t byte_rank[256];                     // id of each single-byte token; a byte that is no token gets a pseudo id
    // Rank maps without all 256 single bytes (the reference accepts any map and fails at ENCODE time, on a piece whose merge
    // leaves such a byte alone: TokenEncoder.java:66-68): every missing byte gets a pseudo id above the table's ids, the merge
    // treats it as any other part, and a document whose output holds one is reported (JTK_ERR_UNENCODABLE), never emitted.
    uint32_t pseudo_base = 0;                    // first pseudo id (0: the table has all 256 bytes)
    int n_missing = 0;
    std::vector<JtkTok8Slot> tok8;               // whole-piece table, pieces of <= 8 bytes
    uint32_t tok8_bits = 0;
    int64_t n_tok8 = 0;
    std::vector<JtkTok16Slot> tok16;             // whole-piece table, pieces of 9..16 bytes
    uint32_t tok16_n = 0;
    int64_t n_tok16 = 0;
    std::vector<uint32_t> bp_rank;               // [65536] rank of the 2-byte token (b0 << 8 | b1), or NONE
    // the same table compressed for LDS: membership bitmap, per-word running count, ranks in index order
    std::vector<uint32_t> pair_in_token;         // [2048] bit (b0 << 8 | b1): the two bytes are adjacent inside some table entry
    std::vector<uint64_t> bp_bits;               // [1024]
    std::vector<uint16_t> bp_cum;                // [1024]
    std::vector<uint32_t> bp_ranks;              // [n_bp] (padded to JTK_BP_MAX)
    std::vector<JtkPairBucket> pair_buckets;     // two-choice cuckoo, primary first: (left,right) -> rank table
    uint32_t pair_bits = 0;
    int64_t n_pairs = 0;
    int64_t pair_displaced = 0, tok8_displaced = 0;   // entries living in their secondary bucket
    // table entries that bytePairMerge of their own bytes does not reproduce (hand-made tables): the whole-piece lookup is
    // then more than a shortcut.  <= 16 bytes: the tok8 / tok16 tables cover them; longer ones are listed in long_tok.
    int64_t n_unreproducible = 0;
    std::vector<JtkLongTokSlot> long_tok;
    std::string long_blob;
    uint32_t long_max_len = 0;
    int64_t n_tokens = 0;
    uint32_t max_id = 0;
};

// Parses `.tiktoken` bytes and builds every derived table.  Returns a jtk_status; `err` gets a message.
int jtk_build_tables(const char* name, int kind, const uint8_t* tiktoken, size_t len,
                     const char* const* special_literals, const int32_t* special_ids, int n_specials,
                     JtkHostTables& out, std::string& err);

// Host decode (Encoding.decodeBytes): returns byte count or a negative status.
int64_t jtk_host_decode(const JtkHostTables& t, const int32_t* ids, int64_t n, uint8_t* out, int64_t cap);

#endif
