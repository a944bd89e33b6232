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
	/** jtk_batch is single-threaded; Encoding must be thread-safe (EncodingRegistry.java:51,61). */
	private final ThreadLocal<Long> batch;

	private HipEncoding(final String name, final int patternKind, final byte[] tiktoken,
			final String[] specialLiterals, final int[] specialIds, final int device) {
		this.name = name;
		this.encodingHandle = nativeCreate(name, patternKind, tiktoken, specialLiterals, specialIds, device);
		this.batch = ThreadLocal.withInitial(() -> nativeBatchCreate(encodingHandle));
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
		return encodeInternal(text, 0, maxTokens);
	}

	@Override
	public List<Integer> encodeOrdinary(final String text) {
		return encodeInternal(text, 1, -1).getTokens();
	}

	@Override
	public EncodingResult encodeOrdinary(final String text, final int maxTokens) {
		return encodeInternal(text, 1, maxTokens);
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
		return nativeDecode(encodingHandle, ids);   // JTK_ERR_UNKNOWN_TOKEN -> IllegalArgumentException
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
		final int[] ids = nativeEncode(batch.get(), utf8, flags, maxTokens, truncated);
		final List<Integer> out = new ArrayList<>(ids.length);
		for (final int id : ids) {
			out.add(id);
		}
		return new EncodingResult(out, truncated[0]);
	}

	// ---- batch: the reason to have a GPU behind the interface ----------------------------------------

	/**
	 * Encodes all documents in one device pass.  {@code utf8} holds the documents' UTF-8 bytes back to
	 * back (direct buffer), {@code docOff} n+1 offsets; returns packed ids plus n+1 token offsets.
	 */
	public BatchResult encodeBatch(final ByteBuffer utf8, final long[] docOff, final boolean ordinary) {
		return nativeEncodeBatch(batch.get(), utf8, docOff, ordinary ? 1 : 0);
	}

	public BatchResult encodeBatch(final List<String> texts, final boolean ordinary) {
		final byte[][] bs = new byte[texts.size()][];
		final long[] off = new long[texts.size() + 1];
		for (int i = 0; i < bs.length; i++) {
			bs[i] = texts.get(i) == null ? new byte[0] : texts.get(i).getBytes(StandardCharsets.UTF_8);
			off[i + 1] = off[i] + bs[i].length;
		}
		final ByteBuffer buf = ByteBuffer.allocateDirect((int) off[bs.length]).order(ByteOrder.nativeOrder());
		for (final byte[] b : bs) {
			buf.put(b);
		}
		buf.flip();
		return encodeBatch(buf, off, ordinary);
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

	@Override
	public void close() {
		nativeDestroy(encodingHandle);
	}

	private static native long nativeCreate(String name, int patternKind, byte[] tiktoken, String[] specialLiterals,
			int[] specialIds, int device);
	private static native void nativeDestroy(long encoding);
	private static native long nativeBatchCreate(long encoding);
	private static native int[] nativeEncode(long batch, byte[] utf8, int flags, int maxTokens, boolean[] truncated);
	private static native BatchResult nativeEncodeBatch(long batch, ByteBuffer utf8, long[] docOff, int flags);
	private static native byte[] nativeDecode(long encoding, int[] ids);

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
