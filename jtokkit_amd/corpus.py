"""Deterministic synthetic corpora for the BASELINE.json configs.

Nothing from the reference travels to the GPU box, and its benchmark corpus (400 Gutenberg books,
benchmark/README.md:9-12) is not in the repository, so workloads are generated here from a seed:
  english(n_docs, mean_bytes)  cfg 1/2/4: Zipf-distributed lower-case words, sentence case, . ? ! , ;
                               contractions, 1-6 digit numbers, paragraphs (\\n\\n), tabs
  mixed(n_docs, mean_bytes)    cfg 3/5: per-document script mix -- English 40 %, CJK/Kana/Hangul 25 %,
                               Cyrillic/Greek/Arabic/Devanagari 15 %, emoji-rich 10 %, code-like 10 %
All letters/digits come from blocks stable since Unicode <= 6.0, so the split does not depend on the
JVM's Unicode version (hard part 3).  Returns (text uint8[n_bytes], doc_off int64[n_docs+1]).
Generation is vectorised numpy: ~1 s per 50 MB.
"""
import numpy as np

_ONSETS = ["", "b", "c", "d", "f", "g", "h", "j", "k", "l", "m", "n", "p", "r", "s", "t", "v", "w", "st", "tr", "ch",
           "sh", "th", "pl", "br", "cr", "gr", "pr", "qu", "sp"]
_NUCLEI = ["a", "e", "i", "o", "u", "ea", "ou", "ai", "io", "ee", "oo", "ie"]
_CODAS = ["", "", "n", "r", "s", "t", "l", "d", "m", "ng", "st", "nt", "ck", "ll", "ss", "rd", "ly", "er", "ed", "es"]
_COMMON = ["the", "of", "and", "to", "a", "in", "is", "that", "it", "was", "for", "on", "with", "as", "be", "at",
           "by", "this", "had", "not", "are", "but", "from", "or", "have", "an", "they", "which", "one", "you",
           "were", "her", "all", "she", "there", "would", "their", "we", "him", "been", "has", "when", "who",
           "will", "more", "no", "if", "out", "so", "said", "what", "up", "its", "about", "into", "than", "them",
           "can", "only", "other", "new", "some", "could", "time", "these", "two", "may", "then", "do", "first",
           "any", "my", "now", "such", "like", "our", "over", "man", "me", "even", "most", "made", "after", "also",
           "did", "many", "before", "must", "through", "years", "where", "much", "your", "way", "well", "down",
           "should", "because", "each", "just", "those", "people", "how", "too", "little", "state", "good", "very",
           "make", "world", "still", "own", "see", "men", "work", "long", "get", "here", "between", "both", "life",
           "being", "under", "never", "day", "same", "another", "know", "while", "last", "might", "us", "great",
           "old", "year", "off", "come", "since", "against", "go", "came", "right", "used", "take", "three"]


_MORE = (
    "time year people way day man thing woman life child world school state family student group country problem "
    "hand part place case week company system program question work government number night point home water room "
    "mother area money story fact month lot right study book eye job word business issue side kind head house "
    "service friend father power hour game line end member law car city community name president team minute idea "
    "kid body information back parent face others level office door health person art war history party result "
    "change morning reason research girl guy moment air teacher force education foot boy age policy music market "
    "sense nation plan college interest death experience effect use class control field development role effort "
    "rate heart drug show leader light voice wife police mind price report decision son view relationship town road "
    "arm difference value building action season society tax director position player record paper space ground "
    "form event official matter center couple site project activity star table need court oil situation cost "
    "industry figure street image phone data picture practice piece land product doctor wall patient worker news "
    "test movie north love step film tree tell ask seem feel try leave call keep let begin help talk turn start "
    "show hear play run move live believe hold bring happen write provide sit stand lose pay meet include continue "
    "set learn lead understand watch follow stop create speak read allow add spend grow open walk win offer "
    "remember consider appear buy wait serve die send expect build stay fall cut reach kill remain suggest raise "
    "pass sell require decide pull return explain hope develop carry break receive agree support hit produce eat "
    "cover catch draw choose important large small different young national political social public possible early "
    "able human local late hard major better economic strong free true full special easy clear recent certain "
    "personal open red difficult available likely short single medical current wrong private past foreign fine "
    "common poor natural significant similar hot dead central happy serious ready simple left physical general "
    "environmental financial blue democratic dark various entire close legal religious cold final main green nice "
    "huge popular traditional cultural always often however again together already almost enough quite rather "
    "really perhaps sometimes usually probably finally especially actually certainly simply nearly quickly "
    "tokenization language model computer science network memory function number string between without during "
    "something nothing everything anyone someone everyone another around though although while until within").split()


def _word_list(rng, n=2048):
    words = list(dict.fromkeys(_COMMON + _MORE))
    seen = set(words)
    while len(words) < n:
        k = int(rng.integers(1, 4))
        w = "".join(_ONSETS[int(rng.integers(len(_ONSETS)))] + _NUCLEI[int(rng.integers(len(_NUCLEI)))]
                    + (_CODAS[int(rng.integers(len(_CODAS)))] if j == k - 1 or rng.random() < 0.3 else "")
                    for j in range(k))
        if w not in seen:
            seen.add(w)
            words.append(w)
    return words


class _Vocab:
    def __init__(self, items):
        bs = [s.encode("utf-8") if isinstance(s, str) else s for s in items]
        self.lens = np.array([len(b) for b in bs], dtype=np.int64)
        self.off = np.zeros(len(bs) + 1, dtype=np.int64)
        np.cumsum(self.lens, out=self.off[1:])
        self.flat = np.frombuffer(b"".join(bs), dtype=np.uint8)

    def __len__(self):
        return len(self.lens)


def _gather(vocab, ids):
    """Concatenate vocab entries ids[0], ids[1], ... -> (bytes, end offset of each entry)."""
    lens = vocab.lens[ids]
    ends = np.cumsum(lens)
    total = int(ends[-1]) if len(ends) else 0
    starts = ends - lens
    idx = np.arange(total, dtype=np.int64)
    idx -= np.repeat(starts, lens)
    idx += np.repeat(vocab.off[ids], lens)
    return vocab.flat[idx], ends


def _zipf_p(n, s=1.1):
    p = 1.0 / np.arange(1, n + 1) ** s
    return p / p.sum()


def _doc_lengths(rng, n_docs, mean, lo, hi, sigma=0.5):
    mu = np.log(mean) - sigma * sigma / 2
    return np.clip(rng.lognormal(mu, sigma, n_docs), lo, hi).astype(np.int64)


def _cut_docs(ends, targets):
    """Cut a stream of entries (end offsets `ends`) into docs of about `targets` bytes, at entry ends."""
    want = np.cumsum(targets)
    k = np.searchsorted(ends, want, side="left")
    k = np.minimum(k, len(ends) - 1)
    k = np.maximum.accumulate(k)
    cuts = ends[k]
    doc_off = np.concatenate([[0], cuts]).astype(np.int64)
    return doc_off


def _stream(rng, vocab_items, probs_fn, target_bytes, mean_entry):
    vocab = _Vocab(vocab_items)
    n_entries = int(target_bytes / mean_entry * 1.15) + 64
    ids = probs_fn(rng, n_entries)
    return _gather(vocab, ids)


def _english_stream(rng, target_bytes, plain=False):
    words = _word_list(np.random.default_rng(12345))
    nw = len(words)
    caps = [w.capitalize() for w in words]
    contr = [w + s for w in words[:256] for s in ("'s", "n't", "'re", "'ve", "'m", "'ll", "'d")]
    nums = [str(int(x)) for x in np.random.default_rng(777).integers(0, 10 ** np.random.default_rng(778).integers(1, 7, 512))]
    if plain:
        seps = [" ", " ", " ", " ", ", ", ". ", "? ", "! "]
        sep_p = np.array([0.2125] * 4 + [0.05, 0.07, 0.015, 0.015])
    else:
        seps = [" ", ", ", ". ", "? ", "! ", ".\n\n", ":\n\t", "; ", " (", ") ", " - ", "...", "\n", "  ", '" ', ' "']
        sep_p = np.array([0.76, 0.06, 0.07, 0.008, 0.008, 0.02, 0.008, 0.008, 0.008, 0.008, 0.008, 0.004, 0.01, 0.004,
                          0.008, 0.008])
    sep_p = sep_p / sep_p.sum()
    items = words + caps + contr + nums + seps
    o_caps, o_contr, o_nums, o_seps = nw, 2 * nw, 2 * nw + len(contr), 2 * nw + len(contr) + len(nums)
    vocab = _Vocab(items)
    zp = _zipf_p(nw)
    n_pairs = int(target_bytes / 6.2 * 1.2) + 64
    w = rng.choice(nw, size=n_pairs, p=zp)
    sep = rng.choice(len(seps), size=n_pairs, p=sep_p)
    kind = rng.random(n_pairs)
    ids_w = w.copy()
    # sentence case: capitalise after a sentence-ending separator
    ender = np.isin(sep, [i for i, s in enumerate(seps) if s[0] in ".?!"])
    after = np.concatenate([[True], ender[:-1]])
    ids_w = np.where(after, w + o_caps, ids_w)
    ids_w = np.where((kind < 0.05) & ~after, o_contr + (w % 256) * 7 + rng.integers(0, 7, n_pairs), ids_w)
    if not plain:
        ids_w = np.where((kind > 0.97) & ~after, o_nums + rng.integers(0, len(nums), n_pairs), ids_w)
    ids = np.empty(2 * n_pairs, dtype=np.int64)
    ids[0::2] = ids_w
    ids[1::2] = sep + o_seps
    data, ends = _gather(vocab, ids)
    return data, ends[1::2]            # documents may end after a separator


def _cjk_stream(rng, target_bytes):
    han = [chr(c) for c in range(0x4E00, 0x4E00 + 3000)]
    hira = [chr(c) for c in range(0x3041, 0x3097)]
    kata = [chr(c) for c in range(0x30A1, 0x30FB)]
    hang = [chr(c) for c in range(0xAC00, 0xAC00 + 2000)]
    punct = ["。", "、", "！", "？", "「", "」", "\n", " "]
    items = han + hira + kata + hang + punct
    vocab = _Vocab(items)
    n = int(target_bytes / 3 * 1.2) + 64
    script = rng.random(n)
    ids = np.where(script < 0.5, rng.choice(len(han), size=n, p=_zipf_p(len(han), 0.9)),
                   np.where(script < 0.7, len(han) + rng.integers(0, len(hira), n),
                            np.where(script < 0.8, len(han) + len(hira) + rng.integers(0, len(kata), n),
                                     len(han) + len(hira) + len(kata) + rng.integers(0, len(hang), n))))
    pun = rng.random(n) < 0.08
    ids = np.where(pun, len(items) - len(punct) + rng.integers(0, len(punct), n), ids)
    data, ends = _gather(vocab, ids)
    return data, ends


def _script_words(rng_seed, ranges, n=1500):
    r = np.random.default_rng(rng_seed)
    out = []
    for _ in range(n):
        lo, hi = ranges[int(r.integers(len(ranges)))]
        k = int(r.integers(2, 9))
        out.append("".join(chr(int(c)) for c in r.integers(lo, hi, k)))
    return out


def _multiscript_stream(rng, target_bytes):
    words = (_script_words(1, [(0x430, 0x450)]) + _script_words(2, [(0x3B1, 0x3C9)]) +
             _script_words(3, [(0x627, 0x63A), (0x641, 0x64A)]) + _script_words(4, [(0x915, 0x939)]))
    seps = [" ", " ", " ", ", ", ". ", "\n", " — ", "! "]
    items = words + seps
    vocab = _Vocab(items)
    n = int(target_bytes / 11 * 1.2) + 64
    block = rng.integers(0, 4, n // 64 + 1).repeat(64)[:n]          # stay in one script for a while
    w = block * 1500 + rng.choice(1500, size=n, p=_zipf_p(1500))
    ids = np.empty(2 * n, dtype=np.int64)
    ids[0::2] = w
    ids[1::2] = len(words) + rng.integers(0, len(seps), n)
    data, ends = _gather(vocab, ids)
    return data, ends[1::2]


def _emoji_stream(rng, target_bytes):
    words = _word_list(np.random.default_rng(12345))[:512]
    emoji = [chr(c) for c in range(0x1F300, 0x1F650)]
    seqs = ["\U0001F468‍\U0001F469‍\U0001F467", "❤️", "\U0001F44D\U0001F3FD", "☺️",
            "\U0001F3F3️‍\U0001F308", "☃️", "\U0001F469‍\U0001F4BB"]
    seps = [" ", " ", "! ", " ", "\n"]
    items = words + emoji + seqs + seps
    vocab = _Vocab(items)
    n = int(target_bytes / 5 * 1.2) + 64
    kind = rng.random(n)
    ids = np.where(kind < 0.55, rng.choice(512, size=n, p=_zipf_p(512)),
                   np.where(kind < 0.93, 512 + rng.integers(0, len(emoji), n),
                            512 + len(emoji) + rng.integers(0, len(seqs), n)))
    out = np.empty(2 * n, dtype=np.int64)
    out[0::2] = ids
    out[1::2] = np.where(kind < 0.55, len(items) - len(seps) + rng.integers(0, len(seps), n), len(items) - len(seps))
    data, ends = _gather(vocab, out)
    return data, ends[1::2]


def _code_stream(rng, target_bytes):
    idents = ["i", "j", "x", "y", "n", "len", "tmp", "self", "value", "index", "count", "result", "data", "buf", "ptr",
              "size", "node", "key", "item", "foo", "bar", "getValue", "set_item", "MAX_LEN", "parseInt", "toString"]
    kws = ["if", "else", "for", "while", "return", "int", "void", "const", "static", "def", "class", "import", "new"]
    punct = [" = ", "(", ")", " {", "}", ";", ", ", ".", " == ", " != ", " += ", "[", "]", "->", "::", " < ", " && ",
             "'quoted'", '"str"', "// note", "0x1F", "42", "1000", "3.14", "\r\n", "\r\n    ", "\r\n        ", "\n\t", " "]
    items = idents + kws + punct
    vocab = _Vocab(items)
    n = int(target_bytes / 4 * 1.2) + 64
    kind = rng.random(n)
    ids = np.where(kind < 0.35, rng.integers(0, len(idents), n),
                   np.where(kind < 0.45, len(idents) + rng.integers(0, len(kws), n),
                            len(idents) + len(kws) + rng.integers(0, len(punct), n)))
    return _gather(vocab, ids)


def _assemble(rng, streams, n_docs_each, mean, lo, hi):
    """Cut each stream into docs, then interleave all docs in a seeded random order."""
    datas, offs = [], []
    for (data, ends), nd in zip(streams, n_docs_each):
        if nd == 0:
            continue
        targets = _doc_lengths(rng, nd, mean, lo, hi)
        doc_off = _cut_docs(ends, targets)
        datas.append(data)
        offs.append(doc_off)
    if len(datas) == 1:
        d = datas[0][:offs[0][-1]]
        return np.ascontiguousarray(d), offs[0]
    base = 0
    starts, lens = [], []
    for data, doc_off in zip(datas, offs):
        starts.append(doc_off[:-1] + base)
        lens.append(np.diff(doc_off))
        base += len(data)
    flat = np.concatenate(datas)
    starts = np.concatenate(starts)
    lens = np.concatenate(lens)
    perm = rng.permutation(len(starts))
    starts, lens = starts[perm], lens[perm]
    ends = np.cumsum(lens)
    total = int(ends[-1])
    idx = np.arange(total, dtype=np.int64)
    idx -= np.repeat(ends - lens, lens)
    idx += np.repeat(starts, lens)
    doc_off = np.concatenate([[0], ends]).astype(np.int64)
    return flat[idx], doc_off


def english(n_docs, mean_bytes=1024, lo=64, hi=8192, seed=2, plain=False):
    rng = np.random.default_rng(seed)
    target = int(n_docs * mean_bytes * 1.05) + 4 * hi
    stream = _english_stream(rng, target, plain=plain)
    return _assemble(rng, [stream], [n_docs], mean_bytes, lo, hi)


def sentences(n_docs=1000, seed=1):
    """cfg 1: short ASCII sentences, 40-120 bytes."""
    return english(n_docs, mean_bytes=80, lo=40, hi=120, seed=seed, plain=True)


def mixed(n_docs, mean_bytes=4096, lo=256, hi=32768, seed=3):
    rng = np.random.default_rng(seed)
    shares = [0.40, 0.25, 0.15, 0.10, 0.10]
    counts = [int(n_docs * s) for s in shares]
    counts[0] += n_docs - sum(counts)
    gens = [_english_stream, _cjk_stream, _multiscript_stream, _emoji_stream, _code_stream]
    streams = []
    for g, c in zip(gens, counts):
        streams.append(g(rng, int(c * mean_bytes * 1.05) + 4 * hi) if c else (np.zeros(0, np.uint8), np.zeros(0, np.int64)))
    return _assemble(rng, streams, counts, mean_bytes, lo, hi)


from .sharding import shard_by_bytes  # noqa: E402,F401  (kept here for callers of corpus.*)
