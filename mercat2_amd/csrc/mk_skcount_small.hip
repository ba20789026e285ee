// mk_skcount_small.hip -- the count stage of mk_skcount.hip with 512-thread workgroups and 4096-slot LDS tables
// (68-76 KB of LDS: two workgroups per CU, or one beside a scatter workgroup of the other context), for chunks cut into
// up to 2^14 buckets.  Experiment of round 4, MK_CORES=1.
#define SKC_SMALL 1
#define SKC_THREADS 512
#define SKC_SLOTS 4096
#define SKC_LB 1024      // (register budget of 4 waves per SIMD: two workgroups of 8 waves per CU)
#define SKC_WGS 2
#define SKC_PRE 2
#define SKC_PUSH 2
#include "mk_skcount.hip"
