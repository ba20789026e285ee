import sys; sys.path.insert(0,'.')
import numpy as np, gzip
from mercat2_amd import native
from oracle import c_oracle
data = gzip.open('tests/golden/inputs/RW1.fna.gz','rb').read()
for k in (33, 63):
    with native.Counter(k) as ctx:
        ctx.count_chunk(data, 1)
        km, cn = ctx.export()
        print(k, ctx.stats()['mode_name'], km.shape, 'exotic', ctx.stats()['exotic_windows'])
    okm, ocn = c_oracle.count(data, k, 1)
    print('rows', km.shape[0], okm.shape[0], 'equal', np.array_equal(km, okm), np.array_equal(cn, ocn))
    a = km.view('S%d' % k).reshape(-1); b = okm.view('S%d' % k).reshape(-1)
    bad = np.flatnonzero(a != b)
    print('mismatches', bad.size, bad[:10])
    if bad.size:
        i = bad[0]
        for j in range(max(0,i-2), i+4): print(j, a[j], cn[j], '|', b[j], ocn[j])
        print('sorted?', np.all(a[:-1] <= a[1:]), 'same set', np.array_equal(np.sort(a), b))
        d = np.flatnonzero(a[:-1] > a[1:]); print('descents', d.size, d[:10])
