// persistent 64-column passes (pcd_prb_kernel, lin_prb_kernel), double storage
#define SPFM_TU_T double
#define SPFM_TU_TAG f64
#include "spfm_engine_prb.inc.h"
