#include "hmsg_common.h"
void hmsg_merge(hmsg_ctx* h) { throw hmsg_error{HMSG_ERR_UNSUPPORTED, "merge not built yet"}; }
