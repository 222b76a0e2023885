// Environment knobs of kernel-tuning experiments.  A product build has none: tuning_env() is a constant null
// there and every use folds to its default.  Build with -DPT_TUNING (make EXTRA=-DPT_TUNING) to get them back.
#pragma once
#include <cstdlib>

namespace pt {

inline const char *tuning_env(const char *name)
{
#ifdef PT_TUNING
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

}  // namespace pt
