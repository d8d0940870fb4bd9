"""Configuration recording (src/configurations.jl:24-79) with the Hubbard compression
`compress = BitArray(conf .== 1)`, `decompress = 2c .- 1` (HubbardModel.jl:56-59).

A compressed configuration is the chunk vector of Julia's BitArray (uint64, element i
-- 1-based, column-major -- in bit (i-1) % 64 of chunk (i-1) / 64) plus its shape; that
is the wire/disk form of the hot path's state (the JLD container itself is out of scope)."""
import numpy as np


class CompressedConf:
    __slots__ = ("chunks", "shape")

    def __init__(self, chunks, shape):
        self.chunks = np.ascontiguousarray(chunks, dtype=np.uint64)
        self.shape = tuple(shape)

    def __eq__(self, other):
        return self.shape == other.shape and np.array_equal(self.chunks, other.chunks)


def compress(conf):
    """host-side compress (the device-side one is DQMC.conf_bits)"""
    c = np.asarray(conf)
    bits = (c.reshape(-1, order="F") == 1).astype(np.uint8)
    pad = (-bits.size) % 64
    bits = np.concatenate([bits, np.zeros(pad, dtype=np.uint8)])
    return CompressedConf(np.packbits(bits, bitorder="little").view(np.uint64), c.shape)


def decompress(cc):
    bits = np.unpackbits(cc.chunks.view(np.uint8), bitorder="little")[: cc.shape[0] * cc.shape[1]]
    return np.asfortranarray((2 * bits.astype(np.int8) - 1).reshape(cc.shape, order="F"))


class ConfigRecorder:
    """ConfigRecorder{CT}(rate): push!(c, mc, model, sweep) keeps every `rate`-th sweep"""

    def __init__(self, rate=10):
        self.configs = []
        self.rate = rate

    def push(self, mc, sweep, walker=0):
        if sweep % self.rate == 0:
            self.configs.append(mc.conf_bits(walker))

    def __len__(self):
        return len(self.configs)

    def __getitem__(self, i):
        return self.configs[i]

    def __iter__(self):
        return iter(self.configs)


class Discarder:
    """Discarder(): drops everything pushed to it"""

    def push(self, *a, **k):
        return None

    def __len__(self):
        return 0

    def __iter__(self):
        return iter(())
