// fh_options.h -- the library's ONE configuration surface.
//
// Everything that selects a code path, sizes a buffer for a test or switches a trace on is a NAMED OPTION of this table:
// set by fh_set_option(name, value) (include/finch_hip.h) or listed in the one environment variable the library reads,
//     FH_DEBUG="name=value,name=value,..."      (a name without "=value" means "1")
// which is looked at in ONE function (fh_options.cpp, refresh_env).  An explicit fh_set_option wins over FH_DEBUG.  No option
// changes a sketch: they choose between exact code paths (A/B measurements, tests that force rare paths on small inputs),
// size thread teams and pools, or print traces.  cfg(name) is what the code asks: the option's value, or nullptr if it is
// not set; an unknown name is a programming error (asserted in debug builds, nullptr otherwise).
#pragma once
#include <stdint.h>

namespace fh {

const char *cfg(const char *name);
inline bool cfg_set(const char *name) { return cfg(name) != nullptr; }
// set, and the value does not begin with '0'
bool cfg_on(const char *name);
uint64_t cfg_u64(const char *name, uint64_t dflt);
// fh_set_option / fh_get_option / fh_option_list
int cfg_assign(const char *name, const char *value); // value == nullptr: back to "not set"; -1: no such option
bool cfg_known(const char *name);
const char *cfg_list();                              // "name\thelp\n" for every option

} // namespace fh
