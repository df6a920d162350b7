// persistent pbcd pass (pbcd_prb_kernel), double storage
#define SPFM_TU_T double
#define SPFM_TU_TAG f64
#include "spfm_engine_pbprb.inc.h"
