package com.knuddels.jtokkit.hip;

import com.knuddels.jtokkit.api.Encoding;
import com.knuddels.jtokkit.api.EncodingResult;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.charset.StandardCharsets;
import java.util.ArrayList;
import java.util.Collections;
import java.util.List;

/**
 * MI355X-backed {@link Encoding}: a thin caller of the C ABI in include/jtokkit_amd.h through the JNI
 * glue in jtk_jni.c.  Register it under a built-in name on a lazy registry before first use and every
 * existing caller of {@code registry.getEncoding(EncodingType.CL100K_BASE)} gets the GPU path:
 *
 * <pre>
 * EncodingRegistry registry = Encodings.newLazyEncodingRegistry();
 * registry.registerCustomEncoding(HipEncoding.cl100kBase(0));   // AbstractEncodingRegistry.java:69-77
 * Encoding enc = registry.getEncoding(EncodingType.CL100K_BASE); // LazyEncodingRegistry.java:18-21
 * </pre>
 *
 * NOT COMPILED in the build image (no JDK there); it documents the binding a maintainer adds.
 * Strings cross as {@code String.getBytes(UTF_8)} (ImmutableByteArray.java:16-19) -- never JNI
 * GetStringUTFChars, whose modified UTF-8 differs for U+0000 and supplementary characters.
 */
public final class HipEncoding implements Encoding, AutoCloseable {

	static {
		System.loadLibrary("jtokkit_amd_jni");
	}

	private final String name;
	private final long encodingHandle;
	/**
	 * Encoding must be thread-safe (EncodingRegistry.java:51,61).  The per-call methods go through ONE jtk_service per
	 * encoding: it is thread-safe and coalesces concurrent callers into device batches.  The batch methods use a
	 * jtk_batch (one caller at a time) taken from a small pool; every batch ever created is tracked so that close()
	 * can destroy them before the encoding.
	 */
	private final long serviceHandle;
	private final java.util.concurrent.ConcurrentLinkedQueue<Long> idleBatches = new java.util.concurrent.ConcurrentLinkedQueue<>();
	private final java.util.Set<Long> allBatches = java.util.concurrent.ConcurrentHashMap.newKeySet();
	private final java.util.regex.Pattern hostPattern;   // null: one of the two patterns the device evaluates
	private volatile boolean closed;

	private HipEncoding(final String name, final int patternKind, final byte[] tiktoken,
			final String[] specialLiterals, final int[] specialIds, final int device, final java.util.regex.Pattern hostPattern) {
		this.name = name;
		this.hostPattern = hostPattern;
		this.encodingHandle = nativeCreate(name, patternKind, tiktoken, specialLiterals, specialIds, device);
		this.serviceHandle = nativeServiceCreate(encodingHandle, 2);
	}

	private HipEncoding(final String name, final int patternKind, final byte[] tiktoken,
			final String[] specialLiterals, final int[] specialIds, final int device) {
		this(name, patternKind, tiktoken, specialLiterals, specialIds, device, null);
	}

	/**
	 * A custom encoding (GptBytePairEncodingParams: name, pattern, mergeable ranks, special tokens) on the device.
	 * Its pattern is matched here on the JVM (java.util.regex, exactly as the reference does) and only the matches go to
	 * the device (jtk_batch_encode_pieces): whole-piece lookup, bytePairMerge and packing.
	 */
	public static HipEncoding custom(final String name, final java.util.regex.Pattern pattern, final byte[] tiktokenFileBytes,
			final String[] specialLiterals, final int[] specialIds, final int device) {
		return new HipEncoding(name, 1, tiktokenFileBytes, specialLiterals, specialIds, device, pattern);
	}

	/**
	 * close() against calls in progress: every method that goes down to the native handles holds the read side for the
	 * duration of the call, close() takes the write side before it destroys them -- it waits for the calls that are inside and
	 * later ones fail with IllegalStateException instead of touching freed memory.
	 */
	private final java.util.concurrent.locks.ReentrantReadWriteLock life = new java.util.concurrent.locks.ReentrantReadWriteLock();

	private void enter() {
		life.readLock().lock();
		if (closed) {
			life.readLock().unlock();
			throw new IllegalStateException("encoding is closed");
		}
	}

	private void leave() {
		life.readLock().unlock();
	}

	private long borrowBatch() {
		enter();
		final Long b = idleBatches.poll();
		if (b != null) {
			return b;
		}
		try {
			final long created = nativeBatchCreate(encodingHandle);
			allBatches.add(created);
			return created;
		} catch (final RuntimeException e) {
			leave();
			throw e;
		}
	}

	private void returnBatch(final long b) {
		idleBatches.add(b);
		leave();
	}

	public static HipEncoding cl100kBase(final int device) {
		return new HipEncoding("cl100k_base", 1, Resources.read("/com/knuddels/jtokkit/cl100k_base.tiktoken"),
				new String[]{"<|endoftext|>", "<|fim_prefix|>", "<|fim_middle|>", "<|fim_suffix|>", "<|endofprompt|>"},
				new int[]{100257, 100258, 100259, 100260, 100276}, device);
	}

	public static HipEncoding r50kBase(final int device) {
		return new HipEncoding("r50k_base", 0, Resources.read("/com/knuddels/jtokkit/r50k_base.tiktoken"),
				new String[]{"<|endoftext|>"}, new int[]{50256}, device);
	}

	public static HipEncoding p50kBase(final int device) {
		return new HipEncoding("p50k_base", 0, Resources.read("/com/knuddels/jtokkit/p50k_base.tiktoken"),
				new String[]{"<|endoftext|>"}, new int[]{50256}, device);
	}

	public static HipEncoding p50kEdit(final int device) {
		return new HipEncoding("p50k_edit", 0, Resources.read("/com/knuddels/jtokkit/p50k_base.tiktoken"),
				new String[]{"<|endoftext|>", "<|fim_prefix|>", "<|fim_middle|>", "<|fim_suffix|>"},
				new int[]{50256, 50281, 50282, 50283}, device);
	}

	// ---- Encoding ------------------------------------------------------------------------------------

	@Override
	public List<Integer> encode(final String text) {
		return encodeInternal(text, 0, -1).getTokens();
	}

	@Override
	public EncodingResult encode(final String text, final int maxTokens) {
		// (a negative maxTokens: the reference's loop never runs -- GptBytePairEncoding.java:79 -- which is what 0 gives: an
		// empty list, truncated iff the text is not empty; -1 is the C ABI's "no limit")
		return encodeInternal(text, 0, Math.max(maxTokens, 0));
	}

	@Override
	public List<Integer> encodeOrdinary(final String text) {
		return encodeInternal(text, 1, -1).getTokens();
	}

	@Override
	public EncodingResult encodeOrdinary(final String text, final int maxTokens) {
		return encodeInternal(text, 1, Math.max(maxTokens, 0));
	}

	// ---- the same without blocking: many documents in flight per thread -----------------------------

	/**
	 * {@link #encode(String)} as a future: the document is handed to the service (jtk_service_submit) and the calling
	 * thread goes on; a daemon thread of this object waits for the tickets in submission order (jtk_service_wait) and
	 * completes the futures.  This is the fast route for the reference's per-call shape
	 * (benchmark/.../AbstractMultiThreadedBenchmark.java:35-45: one task per document): a few threads keep thousands of
	 * documents in flight, so the device batches are large.
	 */
	public java.util.concurrent.CompletableFuture<List<Integer>> encodeAsync(final String text) {
		return submit(text, 0);
	}

	public java.util.concurrent.CompletableFuture<List<Integer>> encodeOrdinaryAsync(final String text) {
		return submit(text, 1);
	}

	private java.util.concurrent.CompletableFuture<List<Integer>> submit(final String text, final int flags) {
		final java.util.concurrent.CompletableFuture<List<Integer>> f = new java.util.concurrent.CompletableFuture<>();
		if (text == null) {
			f.complete(Collections.emptyList());
			return f;
		}
		if (hostPattern != null) {
			f.complete(encodeInternal(text, flags, -1).getTokens());
			return f;
		}
		enter();
		try {
			final long ticket = nativeServiceSubmit(serviceHandle, text.getBytes(StandardCharsets.UTF_8), flags);
			synchronized (inFlight) {
				inFlight.add(new Pending(ticket, f));
				if (completer == null) {
					completer = new Thread(this::completeLoop, "jtokkit-amd-completer");
					completer.setDaemon(true);
					completer.start();
				}
				inFlight.notifyAll();
			}
		} finally {
			leave();
		}
		return f;
	}

	private static final class Pending {
		final long ticket;
		final java.util.concurrent.CompletableFuture<List<Integer>> future;

		Pending(final long ticket, final java.util.concurrent.CompletableFuture<List<Integer>> future) {
			this.ticket = ticket;
			this.future = future;
		}
	}

	private final java.util.ArrayDeque<Pending> inFlight = new java.util.ArrayDeque<>();
	private Thread completer;

	private void completeLoop() {
		for (;;) {
			final Pending p;
			synchronized (inFlight) {
				while (inFlight.isEmpty()) {
					if (closed) {
						return;
					}
					try {
						inFlight.wait(100);
					} catch (final InterruptedException e) {
						return;
					}
				}
			}
			// (the read side is taken BEFORE the ticket leaves the queue: close() waits for an empty queue, then for the write
			// side, so the service outlives every wait)
			life.readLock().lock();
			try {
				synchronized (inFlight) {
					p = inFlight.poll();
				}
				if (p == null) {
					continue;
				}
				try {
					final int[] ids = nativeServiceWait(serviceHandle, p.ticket);      // throws what encode() would
					final List<Integer> out = new ArrayList<>(ids.length);
					for (final int id : ids) {
						out.add(id);
					}
					p.future.complete(out);
				} catch (final RuntimeException e) {
					p.future.completeExceptionally(e);
				}
			} finally {
				life.readLock().unlock();
			}
		}
	}

	@Override
	public int countTokens(final String text) {
		return encode(text).size();
	}

	@Override
	public int countTokensOrdinary(final String text) {
		return encodeOrdinary(text).size();
	}

	@Override
	public String decode(final List<Integer> tokens) {
		return new String(decodeBytes(tokens), StandardCharsets.UTF_8);
	}

	@Override
	public byte[] decodeBytes(final List<Integer> tokens) {
		final int[] ids = new int[tokens.size()];
		for (int i = 0; i < ids.length; i++) {
			ids[i] = tokens.get(i);
		}
		enter();
		try {
			return nativeDecode(encodingHandle, ids);   // JTK_ERR_UNKNOWN_TOKEN -> IllegalArgumentException
		} finally {
			leave();
		}
	}

	@Override
	public String getName() {
		return name;
	}

	private EncodingResult encodeInternal(final String text, final int flags, final int maxTokens) {
		if (text == null) {
			return new EncodingResult(Collections.emptyList(), false);
		}
		final byte[] utf8 = text.getBytes(StandardCharsets.UTF_8);
		final boolean[] truncated = new boolean[1];
		// JTK_ERR_UNSUPPORTED_SPECIAL -> UnsupportedOperationException (thrown by the glue)
		if (hostPattern != null) {
			return encodeBatchPieces(Collections.singletonList(text), (flags & 1) != 0, maxTokens).get(0);
		}
		final int[] ids;
		enter();
		try {
			ids = nativeServiceEncode(serviceHandle, utf8, flags, maxTokens, truncated);
		} finally {
			leave();
		}
		final List<Integer> out = new ArrayList<>(ids.length);
		for (final int id : ids) {
			out.add(id);
		}
		return new EncodingResult(out, truncated[0]);
	}

	// ---- batch: the reason to have a GPU behind the interface ----------------------------------------

	/**
	 * Encodes all documents in one device pass.  {@code utf8} holds the documents' UTF-8 bytes back to
	 * back (direct buffer; best allocated by {@link #allocatePinned(long)}), {@code docOff} n+1 offsets; returns packed ids
	 * plus n+1 token offsets.
	 */
	public BatchResult encodeBatch(final ByteBuffer utf8, final long[] docOff, final boolean ordinary) {
		final long b = borrowBatch();
		try {
			return nativeEncodeBatch(b, utf8, docOff, ordinary ? 1 : 0);
		} finally {
			returnBatch(b);
		}
	}

	public BatchResult encodeBatch(final List<String> texts, final boolean ordinary) {
		final long[] off = new long[texts.size() + 1];
		final ByteBuffer buf = pack(texts, off);
		return encodeBatch(buf, off, ordinary);
	}

	/** Encoding.countTokens / countTokensOrdinary for every text, one device pass, no token ids copied back. */
	public int[] countTokensBatch(final List<String> texts, final boolean ordinary) {
		final long[] off = new long[texts.size() + 1];
		final ByteBuffer buf = pack(texts, off);
		final long b = borrowBatch();
		try {
			final BatchResult r = nativeEncodeBatch(b, buf, off, (ordinary ? 1 : 0) | 4 /* JTK_ENCODE_COUNT_ONLY */);
			final int[] counts = new int[texts.size()];
			for (int d = 0; d < counts.length; d++) {
				throwForStatus(r.status[d]);
				counts[d] = (int) (r.tokOff[d + 1] - r.tokOff[d]);
			}
			return counts;
		} finally {
			returnBatch(b);
		}
	}

	/**
	 * Encoding.encode(text, maxTokens) / encodeOrdinary(text, maxTokens) for every text.  Only the leading bytes of each text are
	 * encoded (jtk_batch_encode_max_tokens), as the reference stops matching at maxTokens (GptBytePairEncoding.java:83-88).
	 */
	public List<EncodingResult> encodeBatch(final List<String> texts, final boolean ordinary, final int maxTokens) {
		if (hostPattern != null) {
			return encodeBatchPieces(texts, ordinary, maxTokens);
		}
		final int limit = Math.max(maxTokens, 0);
		final int n = texts.size();
		final long[] off = new long[n + 1];
		final ByteBuffer buf = pack(texts, off);
		final long b = borrowBatch();
		try {
			// (ids of at most n * limit entries: a limit that makes this exceed an int[] falls back to encoding whole)
			if ((long) n * limit > Integer.MAX_VALUE - 8) {
				final BatchResult r = nativeEncodeBatch(b, buf, off, ordinary ? 1 : 0);
				final long[] kept = new long[n];
				final boolean[] truncated = new boolean[n];
				nativeTruncateBatch(b, limit, kept, truncated);            // GptBytePairEncoding.java:90-100 on the device
				return toResults(r, kept, truncated);
			}
			final int[] ids = new int[n * limit];
			final long[] kept = new long[n];
			final boolean[] truncated = new boolean[n];
			nativeEncodeBatchMaxTokens(b, buf, off, ordinary ? 1 : 0, limit, ids, kept, truncated);
			final List<EncodingResult> out = new ArrayList<>(n);
			for (int d = 0; d < n; d++) {
				final List<Integer> toks = new ArrayList<>((int) kept[d]);
				for (int i = 0; i < kept[d]; i++) {
					toks.add(ids[d * limit + i]);
				}
				out.add(new EncodingResult(toks, truncated[d]));
			}
			return out;
		} finally {
			returnBatch(b);
		}
	}

	/** Encoding.decodeBytes for many token lists in one device pass (special-token ids decode to their literals). */
	public List<byte[]> decodeBytesBatch(final List<List<Integer>> tokenLists) {
		final long[] seqOff = new long[tokenLists.size() + 1];
		for (int q = 0; q < tokenLists.size(); q++) {
			seqOff[q + 1] = seqOff[q] + tokenLists.get(q).size();
		}
		final int[] ids = new int[(int) seqOff[tokenLists.size()]];
		int k = 0;
		for (final List<Integer> l : tokenLists) {
			for (final int id : l) {
				ids[k++] = id;
			}
		}
		final long b = borrowBatch();
		try {
			return java.util.Arrays.asList(nativeDecodeBatch(b, ids, seqOff));   // JTK_ERR_UNKNOWN_TOKEN -> IllegalArgumentException
		} finally {
			returnBatch(b);
		}
	}

	/**
	 * Custom patterns: {@code while (matcher.find())} (GptBytePairEncoding.java:77-80) runs here, its matches go to the
	 * device as byte ranges.  Text between matches is skipped, as in the reference.
	 */
	private List<EncodingResult> encodeBatchPieces(final List<String> texts, final boolean ordinary, final int maxTokens) {
		final long[] off = new long[texts.size() + 1];
		final ByteBuffer buf = pack(texts, off);
		final java.util.ArrayList<Long> begin = new java.util.ArrayList<>();
		final java.util.ArrayList<Long> end = new java.util.ArrayList<>();
		for (int d = 0; d < texts.size(); d++) {
			final String text = texts.get(d) == null ? "" : texts.get(d);
			final java.util.regex.Matcher m = hostPattern.matcher(text);
			long bytePos = off[d];
			int charPos = 0;
			while (m.find()) {
				if (m.end() == m.start()) {
					continue;
				}
				bytePos += text.substring(charPos, m.start()).getBytes(StandardCharsets.UTF_8).length;
				final long len = m.group().getBytes(StandardCharsets.UTF_8).length;
				begin.add(bytePos);
				end.add(bytePos + len);
				bytePos += len;
				charPos = m.end();
			}
		}
		final long[] pb = begin.stream().mapToLong(Long::longValue).toArray();
		final long[] pe = end.stream().mapToLong(Long::longValue).toArray();
		final long b = borrowBatch();
		try {
			final BatchResult r = nativeEncodeBatchPieces(b, buf, off, pb, pe, ordinary ? 1 : 0);
			final long[] kept = new long[texts.size()];
			final boolean[] truncated = new boolean[texts.size()];
			if (maxTokens >= 0) {
				nativeTruncateBatch(b, maxTokens, kept, truncated);
			} else {
				for (int d = 0; d < kept.length; d++) {
					kept[d] = r.tokOff[d + 1] - r.tokOff[d];
				}
			}
			return toResults(r, kept, truncated);
		} finally {
			returnBatch(b);
		}
	}

	/** A direct buffer in page-locked memory (jtk_host_alloc): the device reads it by DMA.  Free with {@link #freePinned}. */
	public static ByteBuffer allocatePinned(final long bytes) {
		return nativeHostAlloc(bytes).order(ByteOrder.nativeOrder());
	}

	public static void freePinned(final ByteBuffer buffer) {
		nativeHostFree(buffer);
	}

	private static ByteBuffer pack(final List<String> texts, final long[] off) {
		final byte[][] bs = new byte[texts.size()][];
		for (int i = 0; i < bs.length; i++) {
			bs[i] = texts.get(i) == null ? new byte[0] : texts.get(i).getBytes(StandardCharsets.UTF_8);
			off[i + 1] = off[i] + bs[i].length;
		}
		final ByteBuffer buf = ByteBuffer.allocateDirect((int) off[bs.length]).order(ByteOrder.nativeOrder());
		for (final byte[] b : bs) {
			buf.put(b);
		}
		buf.flip();
		return buf;
	}

	private static List<EncodingResult> toResults(final BatchResult r, final long[] kept, final boolean[] truncated) {
		final List<EncodingResult> out = new ArrayList<>(kept.length);
		for (int d = 0; d < kept.length; d++) {
			throwForStatus(r.status[d]);
			final List<Integer> ids = new ArrayList<>((int) kept[d]);
			for (long k = r.tokOff[d]; k < r.tokOff[d] + kept[d]; k++) {
				ids.add(r.tokens[(int) k]);
			}
			out.add(new EncodingResult(ids, truncated[d]));
		}
		return out;
	}

	private static void throwForStatus(final int status) {
		if (status == -2) {
			throw new UnsupportedOperationException("Encoding special tokens is not supported yet.");   // GptBytePairEncoding.java:54
		}
		if (status != 0) {
			throw new IllegalStateException("document could not be encoded: jtk_status " + status);
		}
	}

	/** Packed result of a batch: ids of document d are tokens[tokOff[d] .. tokOff[d+1]); status[d] != 0 = error. */
	public static final class BatchResult {
		public final int[] tokens;
		public final long[] tokOff;
		public final int[] status;

		BatchResult(final int[] tokens, final long[] tokOff, final int[] status) {
			this.tokens = tokens;
			this.tokOff = tokOff;
			this.status = status;
		}
	}

	/** Service first (its workers own batches of their own), then every batch this object created, then the encoding. */
	@Override
	public synchronized void close() {
		if (closed) {
			return;
		}
		closed = true;
		synchronized (inFlight) {                                          // (the service's workers finish what is queued before they leave)
			while (!inFlight.isEmpty()) {
				try {
					inFlight.wait(10);
				} catch (final InterruptedException e) {
					break;
				}
			}
		}
		life.writeLock().lock();                                           // calls in progress leave first
		try {
			nativeServiceDestroy(serviceHandle);
			for (final long b : allBatches) {
				nativeBatchDestroy(b);
			}
			allBatches.clear();
			idleBatches.clear();
			nativeDestroy(encodingHandle);
		} finally {
			life.writeLock().unlock();
		}
	}

	private static native long nativeCreate(String name, int patternKind, byte[] tiktoken, String[] specialLiterals,
			int[] specialIds, int device);
	private static native void nativeDestroy(long encoding);
	private static native long nativeServiceCreate(long encoding, int workers);
	private static native void nativeServiceDestroy(long service);
	private static native int[] nativeServiceEncode(long service, byte[] utf8, int flags, int maxTokens, boolean[] truncated);
	private static native long nativeServiceSubmit(long service, byte[] utf8, int flags);
	private static native int[] nativeServiceWait(long service, long ticket);
	private static native long nativeBatchCreate(long encoding);
	private static native void nativeBatchDestroy(long batch);
	private static native BatchResult nativeEncodeBatch(long batch, ByteBuffer utf8, long[] docOff, int flags);
	private static native BatchResult nativeEncodeBatchPieces(long batch, ByteBuffer utf8, long[] docOff, long[] pieceBegin,
			long[] pieceEnd, int flags);
	private static native void nativeTruncateBatch(long batch, long maxTokens, long[] kept, boolean[] truncated);
	private static native void nativeEncodeBatchMaxTokens(long batch, ByteBuffer utf8, long[] docOff, int flags, int maxTokens, int[] ids, long[] kept, boolean[] truncated);
	private static native byte[][] nativeDecodeBatch(long batch, int[] ids, long[] seqOff);
	private static native byte[] nativeDecode(long encoding, int[] ids);
	private static native ByteBuffer nativeHostAlloc(long bytes);
	private static native void nativeHostFree(ByteBuffer buffer);

	private static final class Resources {
		static byte[] read(final String path) {
			try (java.io.InputStream in = HipEncoding.class.getResourceAsStream(path)) {
				if (in == null) {
					throw new IllegalStateException("Could not find " + path + " in resources");
				}
				final java.io.ByteArrayOutputStream out = new java.io.ByteArrayOutputStream();
				final byte[] buf = new byte[1 << 16];
				for (int n; (n = in.read(buf)) > 0; ) {
					out.write(buf, 0, n);
				}
				return out.toByteArray();
			} catch (final java.io.IOException e) {
				throw new IllegalStateException("Could not load " + path + " from resources", e);
			}
		}
	}
}
