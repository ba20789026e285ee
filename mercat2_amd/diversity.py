"""Drop-in for ``compute_alpha_diversity`` of lib/mercat2_diversity.py (lines 13-53).

The reference reads the sample's TSV back and hands the count column to nine scikit-bio functions
(``skbio.diversity.alpha``: shannon, simpson, simpson_e, goods_coverage, fisher_alpha, dominance,
chao1, chao1_ci, ace).  All nine are functions of a few moments of that column -- rows, sum, sum of
squares, sum of c*ln(c) and the number of rows with count 1..10 -- which ``mk_alpha_stats`` reduces on
the GPU from the table that is already there; the closed forms below turn them into the same
numbers, printed the same way (``round(x, 2)``; 'NA' where scikit-bio raises).
"""
from __future__ import annotations

import math
import os
from typing import Dict, Union

from . import native
from .report import _load_tsv

METRICS = ["shannon", "simpson", "simpson_e", "goods_coverage", "fisher_alpha", "dominance", "chao1", "chao1_ci", "ace"]
Z = 1.96            # chao1_ci: scikit-bio's default z-score
RARE = 10           # ace: scikit-bio's default rare_threshold


def _fisher_alpha(n: float, s: float) -> float:
    if s >= n:
        raise RuntimeError("no finite alpha")       # scikit-bio: optimisation fails -> 'NA'
    lo, hi = 1e-12, 1.0
    g = lambda a: a * math.log(1.0 + n / a) - s
    while g(hi) < 0:
        hi *= 2.0
    for _ in range(200):
        mid = 0.5 * (lo + hi)
        if g(mid) < 0:
            lo = mid
        else:
            hi = mid
    return 0.5 * (lo + hi)


def _chao1_ci(n: float, o: int, s: int, d: int):
    if s:
        chao = o + s * (s - 1) / (2.0 * (d + 1))
        if not d:
            var = s * (s - 1) / 2.0 + s * (2 * s - 1) ** 2 / 4.0 - s ** 4 / (4.0 * chao)
        else:
            var = (s * (s - 1) / (2.0 * (d + 1)) + s * (2 * s - 1) ** 2 / (4.0 * (d + 1) ** 2) +
                   s ** 2 * d * (s - 1) ** 2 / (4.0 * (d + 1) ** 4))
        t = chao - o
        k = math.exp(abs(Z) * math.sqrt(math.log(1.0 + var / t ** 2)))
        return o + t / k, o + t * k
    p = math.exp(-n / o)
    half = Z * math.sqrt(o * p / (1 - p))
    low = o / (1 - p) - half
    return (o if o >= low else low), o / (1 - p) + half


def _ace(o: int, freq) -> Union[int, float]:
    s_rare = sum(freq[1:RARE + 1])
    singles = freq[1]
    if singles > 0 and singles == s_rare:
        raise ValueError("all rare species are singletons")
    s_abun = o - s_rare
    if s_rare == 0:
        return s_abun
    n_rare = float(sum(i * freq[i] for i in range(1, RARE + 1)))
    c_ace = 1.0 - singles / n_rare
    top = s_rare * sum(i * (i - 1) * freq[i] for i in range(1, RARE + 1))
    gamma = max(top / (c_ace * n_rare * (n_rare - 1)) - 1.0, 0.0)
    return s_abun + s_rare / c_ace + (singles / c_ace) * gamma


def _fmt(v) -> str:
    if isinstance(v, tuple):
        return "[" + ", ".join(_fmt(x) for x in v) + "]"
    if isinstance(v, int):
        return str(v)
    return repr(round(float(v), 2))


def alpha_from_stats(st: dict) -> Dict[str, str]:
    """{metric: printed value} from the moments returned by ``Counter.alpha_stats()``."""
    o, n, freq = int(st["observed"]), float(st["total"]), list(st["freq"])
    out: Dict[str, str] = {}
    if o == 0:
        return {m: "NA" for m in METRICS}
    dom = st["sum_sq"] / (n * n)
    values = {
        "shannon": lambda: (math.log(n) - st["sum_clnc"] / n) / math.log(2),
        "simpson": lambda: 1.0 - dom,
        "simpson_e": lambda: (1.0 / dom) / o,
        "goods_coverage": lambda: 1.0 - freq[1] / n,
        "fisher_alpha": lambda: _fisher_alpha(n, float(o)),
        "dominance": lambda: dom,
        "chao1": lambda: o + freq[1] * (freq[1] - 1) / (2.0 * (freq[2] + 1)),
        "chao1_ci": lambda: _chao1_ci(n, o, freq[1], freq[2]),
        "ace": lambda: _ace(o, freq),
    }
    for m in METRICS:
        try:
            out[m] = _fmt(values[m]())
        except Exception:
            out[m] = "NA"
    return out


def compute_alpha_diversity(basename: str, counts, out_file, *, device: int = 0) -> Dict[str, str]:
    """compute_alpha_diversity(basename, counts_tsv, out_file) of the reference; ``counts`` may also
    be the sample's Counter (its table is reduced where it is, no TSV re-read)."""
    if isinstance(counts, native.Counter):
        table = alpha_from_stats(counts.alpha_stats())
    else:
        _, kmers, values = _load_tsv(counts)
        with native.Counter(max(1, kmers.shape[1] if kmers.size else 1), native.ALPHABET_RAW, device) as ctx:
            ctx.import_exotic(kmers, values)
            table = alpha_from_stats(ctx.alpha_stats())
    with open(out_file, "w") as w:
        w.write("Metric\t%s\n" % basename)
        for m in METRICS:
            w.write("%s\t%s\n" % (m, table[m]))
    return table
