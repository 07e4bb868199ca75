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



def overlap_distance(a: str, b: str, slack: int = 40) -> int:
    """Edit distance with free end gaps of at most `slack` bases at either end of either sequence: separates END differences (overhangs, trimmed
    tails) from INTERIOR differences of two consensus sequences of the same amplicon.  (Unbounded free ends would make the empty overlap the optimum.)"""
    A = np.frombuffer(a.encode(), dtype=np.uint8); B = np.frombuffer(b.encode(), dtype=np.uint8)
    if len(A) == 0 or len(B) == 0:
        return 0
    BIG = 1 << 28
    idx = np.arange(len(B) + 1, dtype=np.int64)
    prev = np.where(idx <= slack, 0, BIG).astype(np.int64)            # free leading part of b
    best = BIG
    for i in range(1, len(A) + 1):
        sub = prev[:-1] + (B != A[i - 1])
        dele = prev[1:] + 1
        x = np.minimum(sub, dele)
        y = np.concatenate(([0 if i <= slack else BIG], x)) - idx      # free leading part of a
        cur = np.minimum.accumulate(y) + idx
        prev = cur
        if i >= len(A) - slack:
            best = min(best, int(cur[-1]))                             # b consumed, at most `slack` bases of a left over
    return int(min(best, int(prev[max(0, len(B) - slack):].min())))    # a consumed, at most `slack` bases of b left over
