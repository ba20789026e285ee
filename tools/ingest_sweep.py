"""GPU box: where the plain-file window goes under different reader settings: MK_INGEST_BLOCK x reader threads x streams.
   python tools/ingest_sweep.py [reads]"""
import os, sys, time, tempfile, subprocess, json
sys.path.insert(0, ".")
if len(sys.argv) > 2 and sys.argv[1] == "--one":
    from mercat2_amd import native, harness
    path, threads, streams = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    best = None
    for rep in range(3):
        st, tm = {}, {}
        t0 = time.perf_counter()
        harness.run_sample("S2", path, path + ".tsv", 31, 10, 100, streams=streams, threads=threads, stats=st, timings=tm, report=lambda s: None)
        dt = time.perf_counter() - t0
        if best is None or dt < best[0]:
            best = (dt, st, tm)
    dt, st, tm = best
    print(json.dumps({"s": round(dt, 4), "count_file": round(st["s_total"], 4), "retire": round(st["s_retire"], 4), "wait_io": round(st["s_wait_io"], 4),
                      "wait_gpu": round(st["s_wait_gpu"], 4), "feed": round(st["s_feed"], 4), "drain": round(st["s_drain"], 4), "setup": round(st["s_setup"], 4),
                      "tsv": round(tm["tsv_s"], 4), "export": {k: round(v, 4) for k, v in tm["export"].items() if k.startswith("s_")}}))
    sys.exit(0)
from mercat2_amd import native
reads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
d = tempfile.mkdtemp(dir="/tmp")
path = os.path.join(d, "S2.fna")
data = native.synth_reads(10_000_000, 3, reads, 150, 4)
with open(path, "wb") as f:
    f.write(memoryview(data))
del data
for block in (0, 1 << 20, 16 << 20, 64 << 20):
    for threads in (4, 8, 16):
        for streams in (2, 3):
            env = dict(os.environ)
            if block:
                env["MK_INGEST_BLOCK"] = str(block)
            p = subprocess.run([sys.executable, __file__, "--one", path, str(threads), str(streams)], env=env, capture_output=True, text=True)
            print("block=%-9d threads=%-2d streams=%d %s" % (block, threads, streams, p.stdout.strip() or p.stderr[-200:]), flush=True)
