"""mercat2_amd -- MI355X-native k-mer counting behind MerCat2's counting interface.

Only the counting hot path of MerCat2 lives here:

* :func:`mercat2_amd.kmers.find_kmers`   <-> lib/mercat2_kmers.py:32-78
* :class:`mercat2_amd.chunker.Chunker`   <-> lib/mercat2_Chunker.py:14-59
* :func:`mercat2_amd.harness.run_mercat2`, ``chunk_files``, ``countKmers`` <-> bin/mercat2.py:86-137
* ``python -m mercat2_amd.cli``          <-> the -i/-f/-k/-n/-c/-s/-o flags of bin/mercat2.py

All counting is done by hand-written HIP kernels in ``libmercat_hip.so`` (C ABI in
``include/mercat_hip.h``) through ctypes.  There is no CPU fallback: importing works anywhere,
counting raises :class:`mercat2_amd.native.MercatHipError` without a usable HIP device.
"""
__version__ = "0.1.0"
