"""TEST INFRASTRUCTURE ONLY (see oracle/README or DESIGN.md section 7): CPU restatement of the alpha
diversity step, lib/mercat2_diversity.py:13-53.

The arithmetic lives in a third-party dependency that is not in /root/reference and not installed
here: scikit-bio (``skbio.diversity.alpha``; the reference's environment pins scikit-bio 0.5.x).
What follows restates its published formulas, function by function, on a plain list of counts:
shannon (base 2), dominance, simpson, simpson_e (enspie / observed), goods_coverage, fisher_alpha
(the alpha that minimises (alpha*ln(1+N/alpha) - S)^2), chao1 and chao1_ci (bias-corrected, z = 1.96,
with scikit-bio's four variance cases), ace (rare threshold 10).  Pinned by the 98 (sample table,
printed metrics) pairs the reference committed under results/2023-11-29 (tests/golden/diversity/);
those tables hold no singletons or doubletons (all were counted with -c 10), so the singleton and
doubleton branches of chao1_ci / ace are restated from the published formulas without a pin.
"""
import math
from typing import Dict, Iterable, List, Union

METRICS = ["shannon", "simpson", "simpson_e", "goods_coverage", "fisher_alpha", "dominance", "chao1", "chao1_ci", "ace"]


def shannon(counts: List[int]) -> float:
    n = float(sum(counts))
    return -sum((c / n) * math.log(c / n) for c in counts if c) / math.log(2)


def dominance(counts: List[int]) -> float:
    n = float(sum(counts))
    return sum((c / n) ** 2 for c in counts)


def simpson(counts: List[int]) -> float:
    return 1.0 - dominance(counts)


def observed(counts: List[int]) -> int:
    return sum(1 for c in counts if c)


def simpson_e(counts: List[int]) -> float:
    return (1.0 / dominance(counts)) / observed(counts)


def goods_coverage(counts: List[int]) -> float:
    return 1.0 - sum(1 for c in counts if c == 1) / float(sum(counts))


def fisher_alpha(counts: List[int]) -> float:
    n, s = float(sum(counts)), float(observed(counts))
    if s >= n:
        raise RuntimeError("no finite alpha")
    g = lambda a: a * math.log(1.0 + n / a) - s          # increasing in a, from 0 to n
    lo, hi = 1e-12, 1.0
    while g(hi) < 0:
        hi *= 2.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if g(mid) < 0:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


def _osd(counts):
    return observed(counts), sum(1 for c in counts if c == 1), sum(1 for c in counts if c == 2)


def chao1(counts: List[int]) -> float:
    o, s, d = _osd(counts)
    return o + s * (s - 1) / (2.0 * (d + 1))


def _chao1_var(counts) -> float:
    o, s, d = _osd(counts)
    if not d:
        c = chao1(counts)
        return s * (s - 1) / 2.0 + s * (2 * s - 1) ** 2 / 4.0 - s ** 4 / (4.0 * c)
    if not s:
        n = float(sum(counts))
        return o * math.exp(-n / o) * (1 - math.exp(-n / o))
    return (s * (s - 1) / (2.0 * (d + 1)) + s * (2 * s - 1) ** 2 / (4.0 * (d + 1) ** 2) +
            s ** 2 * d * (s - 1) ** 2 / (4.0 * (d + 1) ** 4))


def chao1_ci(counts: List[int], z: float = 1.96):
    o, s, d = _osd(counts)
    if s:
        c = chao1(counts)
        var = _chao1_var(counts)
        t = c - o
        k = math.exp(abs(z) * math.sqrt(math.log(1.0 + var / t ** 2)))
        return o + t / k, o + t * k
    n = float(sum(counts))
    p = math.exp(-n / o)
    return max(o, o / (1 - p) - z * math.sqrt(o * p / (1 - p))), o / (1 - p) + z * math.sqrt(o * p / (1 - p))


def ace(counts: List[int], rare: int = 10) -> Union[int, float]:
    freq = [0] * (rare + 1)
    for c in counts:
        if 1 <= c <= rare:
            freq[c] += 1
    s_rare = sum(freq[1:])
    singles = freq[1]
    if singles > 0 and singles == s_rare:
        raise ValueError("all rare species are singletons")
    s_abun = sum(1 for c in counts if c > rare)
    if s_rare == 0:
        return s_abun
    n_rare = float(sum(i * freq[i] for i in range(1, rare + 1)))
    c_ace = 1.0 - singles / n_rare
    top = s_rare * sum(i * (i - 1) * freq[i] for i in range(1, rare + 1))
    gamma = max(top / (c_ace * n_rare * (n_rare - 1)) - 1.0, 0.0)
    return s_abun + s_rare / c_ace + (singles / c_ace) * gamma


def _fmt(v) -> str:
    """What ``print(func, value)`` shows after the reference's round(x, 2) / [round(x, 2) for x in ...]."""
    if isinstance(v, str):
        return v
    if isinstance(v, tuple):
        return "[" + ", ".join(_fmt(x) for x in v) + "]"
    if isinstance(v, int):
        return str(v)
    return repr(round(float(v), 2))


def alpha_table(counts: Iterable[int]) -> Dict[str, str]:
    """{metric: printed value} exactly as compute_alpha_diversity writes them ('NA' when scikit-bio raises)."""
    counts = [int(c) for c in counts]
    out = {}
    for name in METRICS:
        try:
            out[name] = _fmt(globals()[name](counts))
        except Exception:
            out[name] = "NA"
    return out
