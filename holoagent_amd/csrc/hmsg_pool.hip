#include "hmsg_common.h"
void hmsg_pool(hmsg_ctx* h) { throw hmsg_error{HMSG_ERR_UNSUPPORTED, "pool not built yet"}; }
