// Tier 5 (short_kernel_impl.h) for gap extension 3: a translation unit of its own (compiled in parallel with the others; its code
// object is loaded when a batch with such penalties first takes the tier).
#include "short_kernel_impl.h"

const WfaShortEntry* wfa_short_entry_e3(int ascii, int bt, int l32, int idx) { return ShortTables<3>::pick(ascii, bt, l32, idx); }
