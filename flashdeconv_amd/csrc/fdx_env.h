// Environment switches of libfdx.
//
// RUNTIME switches (the registry in fdx_env.cpp: 25 names, every one exercised by a test or a tool) are read ONCE - when the
// library first asks - and cached; fdx_env_reload() re-reads them (the tests change the environment between fits; a process that
// calls setenv while library threads run must not be raced by getenv at every call).  env("FDX_X") returns the cached value or
// NULL; a name that is not in the registry is a programming error (asserted by tests/test_host.py against the sources).
//
// EXPERIMENT switches - kernel variants and tuning knobs that lost their measurements - exist only in
// builds made with -DFDX_EXPERIMENT (make EXTRA=-DFDX_EXPERIMENT): exp_env() is a constant NULL otherwise and the branches
// behind it fold away.
#pragma once
#include <cstdlib>

namespace fdx {

const char* env(const char* name);
void env_reload();

#ifdef FDX_EXPERIMENT
inline const char* exp_env(const char* name) { return getenv(name); }
#else
constexpr const char* exp_env(const char*) { return nullptr; }
#endif

}  // namespace fdx
