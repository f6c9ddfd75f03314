// attention64b_doff.hip — the per-query data-offset instantiation of attention64b.hip's kernel (attn64b_kernel<true>) as its own
// translation unit, so that the two instantiations compile side by side.
#define ATTN64B_DOFF 1
#include "attention64b.hip"
