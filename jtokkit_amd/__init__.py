"""jtokkit_amd -- MI355X-native batch BPE encode path behind JTokkit's Encoding API."""
from .encoding import Batch, BatchResult, EncodingError, EncodingResult, HipEncoding, HostBuffer, UnsupportedOperationError
from .registry import ENCODING_PARAMS, get_encoding, new_custom_encoding, new_encoding

__all__ = ["Batch", "BatchResult", "HostBuffer", "EncodingError", "EncodingResult", "HipEncoding", "UnsupportedOperationError",
           "ENCODING_PARAMS", "get_encoding", "new_custom_encoding", "new_encoding"]
