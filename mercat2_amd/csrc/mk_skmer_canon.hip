// mk_skmer_canon.hip -- the canonical-key instances (CANON = true) of the super-k-mer partition kernels of mk_skmer.hip,
// compiled as a translation unit of their own (84 of that file's 168 kernel instances: the two halves build side by side).
#define SK_TU_CANON 1
#include "mk_skmer.hip"
