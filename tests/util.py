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


def edit_distance(a, b):
    """Levenshtein distance between two token sequences."""
    a, b = list(a), list(b)
    dp = list(range(len(b) + 1))
    for i in range(1, len(a) + 1):
        prev, dp[0] = dp[0], i
        for j in range(1, len(b) + 1):
            cur = dp[j]
            dp[j] = min(dp[j] + 1, dp[j - 1] + 1, prev + (a[i - 1] != b[j - 1]))
            prev = cur
    return dp[-1]


def token_error_rate(got, want):
    """sum of edit distances / reference tokens over parallel lists of token lists."""
    e = sum(edit_distance(g, w) for g, w in zip(got, want))
    return e / max(sum(len(w) for w in want), 1)
