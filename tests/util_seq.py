import numpy as np


def edit_distance(a: str, b: str) -> int:
    """Levenshtein distance (row-vectorised)."""
    A = np.frombuffer(a.encode(), dtype=np.uint8); B = np.frombuffer(b.encode(), dtype=np.uint8)
    if len(A) == 0 or len(B) == 0:
        return max(len(A), len(B))
    prev = np.arange(len(B) + 1, dtype=np.int32)
    idx = np.arange(len(B) + 1, dtype=np.int32)
    for i in range(1, len(A) + 1):
        sub = prev[:-1] + (B != A[i - 1])
        dele = prev[1:] + 1
        cur = np.empty_like(prev); cur[0] = i
        x = np.minimum(sub, dele)
        # insertion chain: cur[j] = min(x[j-1], cur[j-1]+1)  ->  prefix min of (x - j) + j
        y = np.concatenate(([i], x)) - idx
        cur = np.minimum.accumulate(y) + idx
        prev = cur
    return int(prev[-1])
