// attention64b_doff.hip — the per-query data-offset instantiation of attention64b.hip (its own translation unit: parallel builds)
#define ATTN64B_DOFF 1
#include "attention64b.hip"
