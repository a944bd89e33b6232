package com.knuddels.jtokkit.hip;

import com.knuddels.jtokkit.Encodings;
import com.knuddels.jtokkit.api.EncodingRegistry;

/**
 * Convenience entry point (SURVEY 8 f4): a registry whose four predefined encodings run on an MI355X.
 *
 * <p>The reference's {@code LazyEncodingRegistry} only builds a predefined encoding when its name is still free
 * ({@code LazyEncodingRegistry.java:18-34} via {@code AbstractEncodingRegistry.addEncoding}), so registering the
 * GPU-backed encodings under the reference's names <em>before first use</em> makes every later
 * {@code getEncoding(EncodingType)}, {@code getEncoding(String)} and {@code getEncodingForModel(ModelType)} call return
 * them ({@code AbstractEncodingRegistry.java:22-62}); {@code ModelType} / {@code EncodingType} lookups are untouched.
 * No reference class is edited.
 *
 * <p>Not compiled in this repository's CI (no JDK in the build image); see INTEGRATION.md.
 */
public final class HipEncodings {

	private HipEncodings() {
	}

	/** Same contract as {@code Encodings.newLazyEncodingRegistry()} ({@code Encodings.java:28}), GPU-backed. */
	public static EncodingRegistry newRegistry(final int device) {
		final EncodingRegistry registry = Encodings.newLazyEncodingRegistry();
		registry.registerCustomEncoding(HipEncoding.r50kBase(device));
		registry.registerCustomEncoding(HipEncoding.p50kBase(device));
		registry.registerCustomEncoding(HipEncoding.p50kEdit(device));
		registry.registerCustomEncoding(HipEncoding.cl100kBase(device));
		return registry;
	}

	public static EncodingRegistry newRegistry() {
		return newRegistry(0);
	}
}
