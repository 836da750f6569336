#include "common.h"
thread_local char fva_err_buf[512] = "";
extern "C" const char* fva_last_error(void) { return fva_err_buf; }
extern "C" int fva_version(void) { return 1; }
