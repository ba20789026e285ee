"""TEST INFRASTRUCTURE -- CPU restatement of MerCat2's removeN (lib/mercat2_fasta.py:53-119) and its helper
split_sequenceN (:21-49), in plain Python with the standard library's own textwrap.  Only tests/ may import this;
the product's rewrite is native code (mk_remove_n, mercat2_amd/csrc/mk_host.cpp) and never calls it.

Pinned: tests/test_clean.py checks this restatement against tests/golden/clean_cases.json and tests/golden/clean/*,
which tests/golden/make_clean_golden.py produced by running the reference's own functions in the build container."""
import re
import textwrap
from typing import List, Tuple

_LINE_END = re.compile(r"\r\n|\r|\n")  # text-mode readline: these three end a line


def pieces_of(header: str, sequence: str) -> List[str]:
    """The lines one record with N becomes (lib/mercat2_fasta.py:35-47): cut at every run of N, piece i headed
    '>{first word}_{i} {other words}', its sequence wrapped by textwrap at 80 columns."""
    words = header.split()
    first, rest = words[0], " ".join(words[1:])  # IndexError for an empty header, as in the reference (:40-41)
    out = []
    for number, piece in enumerate(re.split(r"N+", sequence), start=1):
        out.append(">%s_%d %s" % (first, number, rest))
        out.extend(textwrap.wrap(piece, 80))
    return out


def clean_text(text: str, toupper: bool) -> Tuple[str, int, int]:
    """(cleaned text, gc_count, total_length) of lib/mercat2_fasta.py:72-115 for a whole file's decoded text."""
    lines = _LINE_END.split(text)
    if lines and lines[-1] == "":
        lines.pop()
    stripped = [ln.strip() for ln in lines]
    out: List[str] = []
    gc = total = 0
    at = 0
    while at < len(stripped) and not stripped[at].startswith(">"):
        at += 1  # lines in front of the first header are dropped (:74-76, :116)
    while at < len(stripped):
        name = stripped[at][1:]
        at += 1
        body = []
        while at < len(stripped) and not stripped[at].startswith(">"):
            body.append(stripped[at])
            at += 1
        sequence = "".join(body)
        if "N" in sequence:
            for ln in pieces_of(name, sequence):  # :93-104: every line counted, headers too; headers never upper-cased
                out.append(ln if (ln.startswith(">") or not toupper) else ln.upper())
                gc += ln.count("G") + ln.count("C")
                total += len(ln)
        else:
            out.append(">" + name)
            out.extend(ln.upper() if toupper else ln for ln in body)
            gc += sequence.count("G") + sequence.count("C")
            total += len(sequence)
    return "".join(ln + "\n" for ln in out), gc, total
