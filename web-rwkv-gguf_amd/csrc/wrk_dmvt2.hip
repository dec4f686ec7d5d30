// dmv kernels for 2 tokens per launch (see wrk_dmvt_inst.h)
#define DMVT_NT 2
#include "wrk_dmvt_inst.h"
