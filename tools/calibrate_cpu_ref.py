#!/usr/bin/env python3
"""Calibration of bench.py's cpu_baseline: the repo's Python oracle (oracle/cpu_ref.py) against the REFERENCE's
own find_kmers (lib/mercat2_kmers.py, imported by file path) on the same input, one core each.
Run only in the build container (/root/reference mounted):  python tools/calibrate_cpu_ref.py
Writes profiles/round2_calibration.json; bench.py copies the ratio into cpu_baseline.calibration_ratio, so that the
GPU/CPU figure can be read against the reference's speed rather than a stand-in's.
Input = one slice of bench.py's CPU sample: reads 0..39062 of S2 (10 Mbp genome, seeds 3/4), k=31, -c 10."""
import importlib.util
import json
import sys
import tempfile
import time
from pathlib import Path

sys.dont_write_bytecode = True
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mercat2_amd import native  # noqa: E402  (host helper only: the synthetic-read generator)
from oracle import cpu_ref  # noqa: E402


def main():
    spec = importlib.util.spec_from_file_location("ref_kmers", "/root/reference/lib/mercat2_kmers.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    reads, k, c = 39062, 31, 10
    data = native.synth_reads(10_000_000, 3, reads, 150, 4, 0, 0).tobytes()
    with tempfile.TemporaryDirectory() as tmp:
        path = Path(tmp, "slice.fna")
        path.write_bytes(data)
        best = {}
        for name, fn in (("reference", lambda: ref.find_kmers(path, k, c)), ("oracle", lambda: cpu_ref.find_kmers(path, k, c))):
            times, rows = [], None
            for _ in range(3):
                t0 = time.perf_counter()
                table = fn()
                times.append(time.perf_counter() - t0)
                rows = len(table)
            best[name] = (min(times), rows)
    bases = reads * 150
    out = {"sample": "reads 0..%d of S2 (10 Mbp genome, seeds 3/4), k=%d, -c %d, one core, best of 3" % (reads, k, c),
           "bases": bases, "reference_s": best["reference"][0], "oracle_s": best["oracle"][0],
           "reference_bases_per_s": bases / best["reference"][0], "oracle_bases_per_s": bases / best["oracle"][0],
           "ratio_oracle_over_reference": best["reference"][0] / best["oracle"][0],
           "rows": best["oracle"][1], "rows_equal": best["oracle"][1] == best["reference"][1],
           "where": "build container (8-core Xeon 2.1 GHz)"}
    (ROOT / "profiles" / "round2_calibration.json").write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
