"""lzma_amd: MI355X-native LZMA engine behind the API of the Go package itchio/lzma."""
from .api import (BestCompression, BestSpeed, DefaultCompression, ArgumentValueError, HeaderError,
                  StreamError, NewReader, NewWriter, NewWriterLevel, NewWriterSize,
                  NewWriterSizeLevel)
from .codec import (Params, bt4_matches, decode_batch, encode_chunks, make_params,
                    params_from_level)

__all__ = ["BestCompression", "BestSpeed", "DefaultCompression", "ArgumentValueError",
           "HeaderError", "StreamError", "NewReader", "NewWriter", "NewWriterLevel",
           "NewWriterSize", "NewWriterSizeLevel", "Params", "bt4_matches", "decode_batch",
           "encode_chunks", "make_params", "params_from_level"]
