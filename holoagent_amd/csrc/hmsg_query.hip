#include "hmsg_common.h"
struct hmsg_index { std::string err; };
extern "C" {
int hmsg_index_create(int32_t, int32_t, int64_t, const void*, int32_t, const int32_t*, hmsg_index_t** out) { if (out) *out = nullptr; return HMSG_ERR_UNSUPPORTED; }
void hmsg_index_destroy(hmsg_index_t* ix) { delete ix; }
const char* hmsg_index_last_error(const hmsg_index_t* ix) { return ix ? ix->err.c_str() : "null index"; }
int hmsg_query_objects(hmsg_index_t*, int32_t, int32_t, const float*, const int32_t*, const int32_t*, const int32_t*, int32_t, int32_t, int32_t*, int32_t*, double*) { return HMSG_ERR_UNSUPPORTED; }
int hmsg_similarity(hmsg_index_t*, int32_t, const float*, double*) { return HMSG_ERR_UNSUPPORTED; }
}
