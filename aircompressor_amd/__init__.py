"""aircompressor_amd -- MI355X (gfx950) batched block-codec backend for airlift/aircompressor.

Host-side mirror of the reference's `io.airlift.compress.v3` block API over the C ABI of
libaircompressor_hip.so (include/aircompressor_hip.h).  The product path is HIP only: if
the shared library (or a GPU) is missing every codec fails loudly -- there is no CPU fallback.
"""
from .errors import IllegalArgumentException, MalformedInputException, HipUnavailableError
from .native import HipNative, load_library
from .codecs import (
    Compressor,
    Decompressor,
    Lz4HipCompressor,
    Lz4HipDecompressor,
    Lz4FrameHipCompressor,
    Lz4FrameHipDecompressor,
    SnappyFramedHipCompressor,
    Lz4HadoopHipCompressor,
    Lz4HadoopHipDecompressor,
    SnappyHadoopHipCompressor,
    SnappyHadoopHipDecompressor,
    SnappyFramedHipDecompressor,
    SnappyHipCompressor,
    SnappyHipDecompressor,
    ZstdHipCompressor,
    ZstdHipDecompressor,
    ZstdHipInputStream,
    ZstdHipOutputStream,
)
from .xxhash import XxHash32HipHasher, XxHash64HipHasher
from .sharding import shard_for_rank, aggregate_throughput
from .batch import HipBatchCodec, HipMultiContextCodec, OP_LZ4_DECOMPRESS, OP_LZ4_COMPRESS, OP_SNAPPY_DECOMPRESS, OP_SNAPPY_COMPRESS, OP_ZSTD_DECOMPRESS, OP_ZSTD_COMPRESS, OP_LZ4FRAME_DECOMPRESS, OP_LZ4FRAME_COMPRESS, OP_SNAPPYFRAMED_DECOMPRESS, OP_SNAPPYFRAMED_COMPRESS, OP_LZ4HADOOP_DECOMPRESS, OP_LZ4HADOOP_COMPRESS, OP_SNAPPYHADOOP_DECOMPRESS, OP_SNAPPYHADOOP_COMPRESS, OP_ZSTDSTREAM_COMPRESS, partition_blocks

__all__ = [
    "IllegalArgumentException", "MalformedInputException", "HipUnavailableError", "HipNative", "load_library",
    "Compressor", "Decompressor", "Lz4HipCompressor", "Lz4HipDecompressor", "Lz4FrameHipCompressor", "Lz4FrameHipDecompressor", "SnappyFramedHipCompressor", "SnappyFramedHipDecompressor", "Lz4HadoopHipCompressor", "Lz4HadoopHipDecompressor", "SnappyHadoopHipCompressor", "SnappyHadoopHipDecompressor", "SnappyHipCompressor", "SnappyHipDecompressor",
    "ZstdHipCompressor", "ZstdHipDecompressor", "ZstdHipOutputStream", "ZstdHipInputStream", "XxHash32HipHasher", "XxHash64HipHasher", "HipBatchCodec", "HipMultiContextCodec", "partition_blocks", "shard_for_rank", "aggregate_throughput",
    "OP_LZ4_DECOMPRESS", "OP_LZ4_COMPRESS", "OP_SNAPPY_DECOMPRESS", "OP_SNAPPY_COMPRESS", "OP_ZSTD_DECOMPRESS", "OP_ZSTD_COMPRESS", "OP_LZ4FRAME_DECOMPRESS", "OP_LZ4FRAME_COMPRESS", "OP_SNAPPYFRAMED_DECOMPRESS", "OP_SNAPPYFRAMED_COMPRESS", "OP_LZ4HADOOP_DECOMPRESS", "OP_LZ4HADOOP_COMPRESS", "OP_SNAPPYHADOOP_DECOMPRESS", "OP_SNAPPYHADOOP_COMPRESS", "OP_ZSTDSTREAM_COMPRESS",
]
