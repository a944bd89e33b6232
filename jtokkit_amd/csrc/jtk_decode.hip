// jtk_decode.hip -- batch decode on the device: Encoding.decodeBytes(List<Integer>)
// (GptBytePairEncoding.java:137-151: for every token id, append its byte string; unknown id ->
// IllegalArgumentException, :302-314 incl. special tokens :308-311) for many token lists at once.
//
//   dec_mark     one bit per sequence start (token index), like mark_docs
//   dec_count    bytes per tile of 2048 tokens (lengths from the offset table); unknown ids -> status of their sequence
//   dec_scan     exclusive scan of the tile sizes (one workgroup; tiles are few)
//   dec_scatter  per tile: byte offset of every token (block scan), the tile's bytes assembled in LDS and
//                written in aligned 4-byte words; leaves the byte offset at sequence starts
//   dec_offsets  byte_off[q] per sequence
// Integer / byte gather work; bound by the random reads of the token byte strings (L2-resident blob).
#include "jtk_kernels.h"

namespace {

constexpr int DT = JTK_DEC_TILE;           // tokens per tile
constexpr int DSTAGE = 16384;              // bytes of a tile assembled in LDS (ordinary text: ~7 KB)

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)v, d);
        if (lane >= d) v += o;
    }
    return v;
}

__device__ __forceinline__ uint32_t tok_len(const JtkDecodeWork& w, int32_t id) {
    if (id < 0 || (uint32_t)id >= w.n_ids_table) return 0u;
    return w.tab_off[id + 1] - w.tab_off[id];
}

__global__ void __launch_bounds__(256) k_dec_mark(JtkDecodeWork w) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q > w.n_seqs) return;
    const int64_t p = w.seq_off[q];
    if (p < 0 || p > w.n_tok) return;
    atomicOr((unsigned long long*)&w.seqmask[p >> 6], 1ull << (p & 63));
}

__global__ void __launch_bounds__(256) k_dec_count(JtkDecodeWork w) {
    __shared__ uint32_t s_wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t t0 = (int64_t)blockIdx.x * DT + tid * 8;
    uint32_t sum = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int64_t t = t0 + j;
        if (t < w.n_tok) {
            const int32_t id = w.ids[t];
            const uint32_t l = tok_len(w, id);
            if (l == 0) {                                              // GptBytePairEncoding.java:313
                int64_t lo = 0, hi = w.n_seqs;                         // sequence containing token t
                while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (w.seq_off[mid] > t) hi = mid; else lo = mid + 1; }
                if (lo >= 1) atomicMin(&w.status[lo - 1], -3 /* JTK_ERR_UNKNOWN_TOKEN */);
            }
            sum += l;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) sum += (uint32_t)__shfl_xor((int)sum, d);
    if (lane == 0) s_wsum[wv] = sum;
    __syncthreads();
    if (tid == 0) w.tile_bytes[blockIdx.x] = s_wsum[0] + s_wsum[1] + s_wsum[2] + s_wsum[3];
}

__global__ void __launch_bounds__(1024) k_dec_scan(JtkDecodeWork w) {
    __shared__ uint64_t s_wsum[16];
    __shared__ uint64_t s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int64_t c0 = 0; c0 < w.n_tiles; c0 += 4096) {
        const int64_t i0 = c0 + (int64_t)tid * 4;
        uint32_t v[4];
        uint32_t sum = 0;
        for (int j = 0; j < 4; j++) { v[j] = (i0 + j < w.n_tiles) ? w.tile_bytes[i0 + j] : 0u; sum += v[j]; }
        const uint32_t inc = wave_incl_scan_u32(sum);
        if (lane == 63) s_wsum[wv] = inc;
        __syncthreads();
        uint64_t before = s_base;
        for (int k = 0; k < wv; k++) before += s_wsum[k];
        uint64_t run = before + inc - sum;
        for (int j = 0; j < 4; j++) { if (i0 + j < w.n_tiles) w.tile_off[i0 + j] = (int64_t)run; run += v[j]; }
        __syncthreads();
        if (tid == 1023) s_base = run;
        __syncthreads();
    }
    if (tid == 0) { w.tile_off[w.n_tiles] = (int64_t)s_base; *w.total = (int64_t)s_base; }
}

__global__ void __launch_bounds__(256) k_dec_scatter(JtkDecodeWork w) {
    __shared__ __attribute__((aligned(16))) uint8_t s_out[DSTAGE + 8];
    __shared__ uint32_t s_wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t tile = blockIdx.x;
    const int64_t t0 = tile * DT + tid * 8;
    const int64_t obase = w.tile_off[tile];
    const uint32_t total = w.tile_bytes[tile];
    const bool stage = total <= (uint32_t)DSTAGE && w.out != nullptr;
    int32_t id[8];
    uint32_t len[8], sum = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int64_t t = t0 + j;
        id[j] = (t < w.n_tok) ? w.ids[t] : -1;
        len[j] = (t < w.n_tok) ? tok_len(w, id[j]) : 0u;
        sum += len[j];
    }
    const uint32_t inc = wave_incl_scan_u32(sum);
    if (lane == 63) s_wsum[wv] = inc;
    __syncthreads();
    uint32_t pre = inc - sum;
    for (int k = 0; k < wv; k++) pre += s_wsum[k];
    const uint64_t mword = w.seqmask[t0 >> 6];                        // (t0 is a multiple of 8: one word covers the 8 tokens)
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int64_t t = t0 + j;
        if (t < w.n_tok && ((mword >> (t & 63)) & 1ull)) w.seqpre[t] = pre;
        if (len[j] && w.out != nullptr) {
            const uint8_t* src = w.tab_blob + w.tab_off[id[j]];
            if (stage) for (uint32_t i = 0; i < len[j]; i++) s_out[pre + i] = src[i];
            else for (uint32_t i = 0; i < len[j]; i++) w.out[obase + pre + i] = src[i];
        }
        pre += len[j];
    }
    if (!stage) return;
    __syncthreads();
    // aligned 4-byte words of the output that the tile's bytes [obase, obase + total) touch
    const int64_t a0 = obase & ~(int64_t)3, a1 = (obase + total + 3) & ~(int64_t)3;
    for (int64_t g = a0 + (int64_t)tid * 4; g < a1; g += 1024) {
        const int64_t rel = g - obase;                                 // may be -3..-1 for the first word
        if (rel >= 0 && rel + 4 <= (int64_t)total) {
            const uint32_t v = (uint32_t)s_out[rel] | ((uint32_t)s_out[rel + 1] << 8) | ((uint32_t)s_out[rel + 2] << 16) | ((uint32_t)s_out[rel + 3] << 24);
            *reinterpret_cast<uint32_t*>(w.out + g) = v;
        } else {
            for (int k = 0; k < 4; k++) { const int64_t r = rel + k; if (r >= 0 && r < (int64_t)total) w.out[g + k] = s_out[r]; }
        }
    }
}

__global__ void __launch_bounds__(256) k_dec_offsets(JtkDecodeWork w) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q > w.n_seqs) return;
    const int64_t p = w.seq_off[q];
    w.byte_off[q] = (p >= w.n_tok) ? w.tile_off[w.n_tiles] : w.tile_off[p / DT] + w.seqpre[p];
    if (q < w.n_seqs) {
        const int32_t st = w.status[q];
        if (st < 0) atomicMin(w.worst_status, st);
    }
}

// ---------------------------------------------------------------------------------------------------
// truncate: Encoding.encode(text, maxTokens) for every document of the last batch encode
// (GptBytePairEncoding.java:43-45, 79, 90-100, 110-119).  Pieces encode independently, so the list before
// the back-off is the first min(maxTokens, total) tokens of the full result; the back-off (:90-100) drops
// trailing tokens until decode(tokens) is a prefix of the text, i.e. until their bytes end on a code-point
// boundary -- or in the middle of a U+FFFD of the text, which new String(bytes, UTF_8) turns into the
// same U+FFFD.  truncated (:97) = text.length() > decoded.length().  One thread per document.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_truncate(JtkTruncWork w) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= w.n_docs) return;
    const int64_t t0 = w.tok_off[d], cnt = w.tok_off[d + 1] - t0;
    int64_t keep = cnt < w.max_tokens ? cnt : w.max_tokens;
    uint8_t trunc = 0;
    if (cnt > w.max_tokens) {
        const uint8_t* tx = w.text + w.doc_off[d];
        const int64_t len = w.doc_off[d + 1] - w.doc_off[d];
        int64_t nb = 0;
        for (int64_t k = 0; k < keep; k++) { const int32_t id = w.tokens[t0 + k]; nb += w.tab_off[id + 1] - w.tab_off[id]; }
        for (;; keep--) {
            const bool boundary = (nb == len) || ((tx[nb] & 0xC0) != 0x80);
            if (boundary) { trunc = nb < len; break; }
            int64_t c = nb;
            while (c > 0 && (tx[c] & 0xC0) == 0x80) c--;
            if (c + 2 < len && tx[c] == 0xEF && tx[c + 1] == 0xBF && tx[c + 2] == 0xBD) { trunc = c + 3 < len; break; }
            if (keep == 0) break;
            const int32_t id = w.tokens[t0 + keep - 1];
            nb -= w.tab_off[id + 1] - w.tab_off[id];
        }
    }
    w.kept[d] = keep;
    w.truncated[d] = trunc;
}

}  // namespace

void jtk_launch_truncate(const JtkTruncWork& w, hipStream_t s) {
    if (w.n_docs > 0) hipLaunchKernelGGL(k_truncate, dim3((unsigned)((w.n_docs + 255) / 256)), dim3(256), 0, s, w);
}

void jtk_launch_decode_count(const JtkDecodeWork& w, hipStream_t s) {
    const int64_t n = w.n_seqs + 1;
    hipLaunchKernelGGL(k_dec_mark, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w);
    hipLaunchKernelGGL(k_dec_count, dim3((unsigned)w.n_tiles), dim3(256), 0, s, w);
    hipLaunchKernelGGL(k_dec_scan, dim3(1), dim3(1024), 0, s, w);
}
void jtk_launch_decode_scatter(const JtkDecodeWork& w, hipStream_t s) {
    const int64_t n = w.n_seqs + 1;
    hipLaunchKernelGGL(k_dec_scatter, dim3((unsigned)w.n_tiles), dim3(256), 0, s, w);
    hipLaunchKernelGGL(k_dec_offsets, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w);
}
