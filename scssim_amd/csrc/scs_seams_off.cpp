// product build: no seam is ever set (scs_seams.h)
#include "scs_seams.h"
namespace scs { const char* seam_env(const char*) { return nullptr; } }
