// dmv kernels for 4 tokens per launch (see wrk_dmvt_inst.h)
#define DMVT_NT 4
#include "wrk_dmvt_inst.h"
