// attention64b_qk8.hip — the e4m3 Q, K instantiation of attention64b.hip (fixed zero softmax offset)
#define ATTN64B_QK8 1
#include "attention64b.hip"
