import numpy as np


def bf16_round(x):
    """float32 -> nearest bfloat16 (RNE) -> float32, same conversion as the engine."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(x.shape)


def rnd(dtype, x):
    return bf16_round(x) if dtype == 1 else np.ascontiguousarray(x, np.float32)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)
