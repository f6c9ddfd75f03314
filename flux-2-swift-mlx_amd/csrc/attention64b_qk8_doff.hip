// attention64b_qk8_doff.hip — the e4m3 Q, K instantiation of attention64b.hip with the per-query data offset
#define ATTN64B_DOFF 1
#define ATTN64B_QK8 1
#include "attention64b.hip"
