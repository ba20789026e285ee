"""First level of a two-level partition, timed: the product library against a build with -DSK_ABL_COARSE (the queue
scatter files every record under bucket & ~63: p1 / 64 = 128 coarse buckets, 6.5 records per (tile, bucket) run instead
of one -- the tables that come out are garbage, only the kernel's duration under rocprofv3 means anything).
  rocprofv3 --kernel-trace --stats -d out -- python3 tools/coarse_scatter_probe.py      (MERCAT_HIP_LIB picks the build)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from mercat2_amd import native
text = native.synth_reads(10_000_000, 3, 660_000, 150, 4)
buf = torch.from_numpy(text).cuda()
with native.Counter(31, native.ALPHABET_NT2) as ctx:
    for i in range(8):
        try:
            ctx.count_device(buf.data_ptr(), buf.numel(), 10)
        except Exception as e:  # the ablation build's count stage may refuse what the partition handed it
            print("chunk", i, "refused:", str(e)[:100], flush=True)
    torch.cuda.synchronize()
    print(ctx.stats(), flush=True)
