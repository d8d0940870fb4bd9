// The commit the library was built from: short hash, "+" appended when the sources differed from it; "unknown" outside git.
// Recompiled by every make (csrc/Makefile), so that bench.py can mark committed profile files as stale (dqmc_build_commit()).
#include "../../include/dqmc_hip.h"
#ifndef DQMC_BUILD_COMMIT
#define DQMC_BUILD_COMMIT "unknown"
#endif
#ifndef DQMC_SOURCE_HASH
#define DQMC_SOURCE_HASH "unknown"
#endif
extern "C" const char *dqmc_build_commit(void) { return DQMC_BUILD_COMMIT; }
// sha256 (12 hex digits) over csrc/*.hip, *.h, *.inl, engine.cpp, the Makefile and the assembly patcher: identical for two commits
// that build the same kernels
extern "C" const char *dqmc_build_source_hash(void) { return DQMC_SOURCE_HASH; }
